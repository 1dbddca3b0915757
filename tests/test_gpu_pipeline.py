"""GPU tests of the callers either side of the path: odd / large point counts against the oracle, the
inference_demo call surface end to end, the eval harness (per-gender batching, V2V / MPJPE), the stream pipeline."""
import os
import types

import numpy as np
import pytest
import torch

from etch_amd.utils.weights import load_seeded, seeded_state_dict

pytestmark = pytest.mark.gpu


def scan(seed, n):
    return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)


def make(tmp_path, seed=1):
    from etch_amd import constants as K
    from etch_amd.models.models_pointcloud import GT_network_equiv
    from etch_amd.utils.body_model import SyntheticSMPL
    args = types.SimpleNamespace(output_folder=str(tmp_path), EPN_input_radius=0.4, EPN_layer_num=2, device=torch.device("cuda"),
                                 markerset=K.default_markerset(), scale_magnitude=10, body_model=SyntheticSMPL(7))
    return args, load_seeded(GT_network_equiv(option=args), seed).cuda().eval()


@pytest.mark.parametrize("B,N", [(1, 777), (3, 1501)])
def test_odd_point_counts_vs_oracle(tmp_path, B, N):
    """Ragged sizes: N not a multiple of any tile / stride (ceil-divided strides, partial waves, partial GEMM tiles)."""
    from oracle import stage1 as S1
    args, model = make(tmp_path)
    pts = torch.from_numpy(np.stack([scan(50 + b, N) for b in range(B)]))
    with torch.no_grad():
        res, _ = model(pts.cuda(), ["confidence", "direction", "magnitude"], "standard_vector")
    sd = {k: v.cpu() for k, v in seeded_state_dict(model, 1).items()}
    ref = S1.forward(sd, pts, S1.build_layer_table(), return_aux=True)
    for k in ("part_labels", "confidences", "magnitude"):
        assert float((res[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()) < 1e-4, k
    assert float((model.last_anc_w.cpu() - ref["anc_w"]).abs().max() / ref["anc_w"].abs().max()) < 1e-4


def test_inference_demo_cli_end_to_end(tmp_path):
    """inference_demo.main on a synthetic OBJ: output files, npz keys / shapes (inference_demo.py:108-127), un-centring."""
    from etch_amd import inference_demo as D
    rng = np.random.default_rng(0)
    # a closed-ish blob mesh: icosphere-free: random convex hull-less strip is enough for area-weighted sampling
    v = scan(3, 600) + np.array([1.0, 2.0, 3.0], np.float32)
    f = np.stack([np.arange(0, 598), np.arange(1, 599), np.arange(2, 600)], 1)
    obj = tmp_path / "scan_000.obj"
    with open(obj, "w") as fh:
        for p in v:
            fh.write(f"v {p[0]} {p[1]} {p[2]}\n")
        for t in f:
            fh.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
    out = tmp_path / "out"
    # a seeded checkpoint: without --model_path the weights are torch's unseeded default init (as in the reference) and the fit
    # of a random network's output is chaotic, which made this test flaky
    args0, model0 = make(tmp_path / "ckpt_model")
    ckpt = tmp_path / "ckpt.pth"
    torch.save({k: v.cpu() for k, v in model0.state_dict().items()}, ckpt)
    argv = ["--scan_path", str(obj), "--output_folder", str(out), "--num_point", "1024", "--synthetic_body", "--markerset_path", "missing.json",
            "--model_path", str(ckpt)]
    D.main(argv)
    info = np.load(out / "scan_000_output_smpl_info.npz")
    assert {k: info[k].shape for k in info.files} == {"body_pose": (21, 3), "hand_pose": (2, 3), "betas": (10,), "global_orient": (3,),
                                                       "transl": (3,), "joints": (45, 3)}
    mesh = D.load_obj(str(out / "scan_000_pred_smpl.obj"))
    assert mesh.vertices.shape == (6890, 3) and np.isfinite(mesh.vertices).all()
    # the fitted body is moved back by the bbox centre of the scan (inference_demo.py:25-28,108)
    centre = (v.min(0) + v.max(0)) / 2
    assert np.abs(mesh.vertices.mean(0) - centre).max() < 1.0
    assert os.path.exists(out / "EPN_model_setting_json")
    # the CLI result is the library call's result: same sampled points through predict_smpl with the same checkpoint
    centred, c0 = D.preprocess_scan(str(obj))
    np.testing.assert_allclose(c0, centre, atol=1e-6)
    m1, info1 = D.predict_smpl(args0, model0, D.sample_points_from_mesh(centred, 1024, 0))
    np.testing.assert_array_equal(np.asarray(info1[1][0]), info["betas"])
    np.testing.assert_allclose(m1.vertices + c0, mesh.vertices, atol=2e-6)   # OBJ text round trip
    # and it is reproducible run to run
    D.main(argv[:3] + [str(tmp_path / "out2")] + argv[4:])
    info2 = np.load(tmp_path / "out2" / "scan_000_output_smpl_info.npz")
    for k in info.files:
        np.testing.assert_array_equal(info[k], info2[k])


def test_eval_harness_gender_groups_and_metrics(tmp_path):
    from etch_amd import eval as E
    from etch_amd.models.fit_SMPL import fit_smpl
    from etch_amd.utils.body_model import SyntheticSMPL
    args, model = make(tmp_path)
    args.body_model = {"male": SyntheticSMPL(7), "female": SyntheticSMPL(8)}
    B, N = 4, 900
    pts = torch.from_numpy(np.stack([scan(90 + b, N) for b in range(B)])).cuda()
    genders = ["male", "female", "female", "male"]
    gt_v = np.zeros((B, 6890, 3), np.float32)
    gt_j = np.zeros((B, 45, 3), np.float32)
    recs = E.evaluate_batch(args, model, pts, genders, ids=[f"s{i}" for i in range(B)], gt_vertices=gt_v, gt_joints=gt_j,
                            output_folder=str(tmp_path / "eval"))
    assert [r["id"] for r in recs] == ["s0", "s1", "s2", "s3"]
    # per-sample results equal a per-sample call with that sample's gender (eval.py:191-209 semantics)
    with torch.no_grad():
        res, _ = model(pts, ["confidence", "direction", "magnitude"], "standard_vector")
    from etch_amd import ops
    labels = ops.argmax_rows(res["part_labels"])
    inner = ops.inner_points(pts, res["direction"], res["magnitude"], 10.0)
    for j in (1, 3):
        m, _, _, info = fit_smpl(args, inner[j:j + 1].contiguous(), labels[j:j + 1].contiguous(), res["confidences"][j:j + 1].contiguous(), genders[j])
        assert abs(E.v2v(gt_v[j], m[0].vertices) - recs[j]["v2v"]) < 1e-6
        assert abs(E.mpjpe(gt_j[j], info[4][0]) - recs[j]["mpjpe"]) < 1e-6
    assert os.path.exists(tmp_path / "eval" / "v2v_score.txt") and os.path.exists(tmp_path / "eval" / "s2" / "output_smpl_info_s2.npz")
    assert E.v2v(np.ones((5, 3)), np.zeros((5, 3))) == pytest.approx(np.sqrt(3))
    assert E.mpjpe(np.zeros((45, 3)), np.ones((45, 3))) == pytest.approx(np.sqrt(3))


@pytest.mark.parametrize("nb,B,N", [(3, 2, 640), (5, 8, 5000)])
def test_stream_pipeline_matches_synchronous_path(tmp_path, nb, B, N):
    """Two batches in flight, heads joined on the stage-2 stream, index ops of the next batch running early: same bits as one
    batch at a time (the larger case keeps every stream busy long enough for a lifetime / ordering bug to bite; batches are
    dropped by the caller right after submit)."""
    from etch_amd.inference_demo import predict_smpl_batch
    from etch_amd.pipeline import HotPathPipeline
    args, model = make(tmp_path)
    mk = lambda k: torch.from_numpy(np.stack([scan(200 + 10 * k + b, N) for b in range(B)])).cuda()
    ref = [predict_smpl_batch(args, model, mk(k), "neutral") for k in range(nb)]
    pipe = HotPathPipeline(args, model, "neutral", max_in_flight=2)
    got = list(pipe.run(mk(k) for k in range(nb)))
    # NaN-aware: with seeded random weights some confidences are ~0.005, conf**20 underflows to 0 and the weighted marker centre is
    # 0/0 -- in the reference as well (fit_SMPL.py:52-57); NaN markers must then be NaN in both schedules
    same = lambda a, b: np.array_equal(np.asarray(a.cpu() if torch.is_tensor(a) else a), np.asarray(b.cpu() if torch.is_tensor(b) else b), equal_nan=True)
    for (m0, mk0, v0, i0), (m1, mk1, v1, i1) in zip(ref, got):
        assert same(mk0, mk1) and same(v0, v1)
        for a, b in zip(i0, i1):
            assert same(a, b)
        assert same(m0[0].vertices, m1[0].vertices)


@pytest.mark.parametrize("B,N", [(2, 1500), (8, 5000)])
def test_schedule_variants_are_bitwise_identical(tmp_path, B, N):
    """Stream schedule must not change a bit: heads on their own streams vs one stream, index ops prefetched on the side
    stream vs computed inline, repeated to give a cross-stream race the chance to show."""
    args, model = make(tmp_path)
    pts = torch.from_numpy(np.stack([scan(400 + b, N) for b in range(B)])).cuda()
    items = ["confidence", "direction", "magnitude"]

    def run(concurrent, overlap):
        model.concurrent_heads, model.overlap_index_ops = concurrent, overlap
        with torch.no_grad():
            res, _ = model(pts, items, "standard_vector")
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in res.items()}
    try:
        ref = run(False, False)
        for rep in range(4):
            for concurrent, overlap in ((True, True), (False, True), (True, False)):
                got = run(concurrent, overlap)
                for k in ref:
                    assert torch.equal(ref[k], got[k]), (k, concurrent, overlap, rep)
    finally:
        model.concurrent_heads, model.overlap_index_ops = type(model).concurrent_heads, type(model).overlap_index_ops


def test_pred_item_subsets_match_full_forward(tmp_path):
    """forward(pred_items=...) with any subset of the heads (the reference's default is ["direction", "magnitude"],
    models_pointcloud.py:146) returns exactly the entries of the full forward: no head depends on another having run."""
    args, model = make(tmp_path)
    pts = torch.from_numpy(np.stack([scan(500 + b, 1200) for b in range(2)])).cuda()
    with torch.no_grad():
        full, idx_full = model(pts, ["confidence", "direction", "magnitude"], "standard_vector")
        for items in (["direction", "magnitude"], ["direction"], ["confidence"], ["magnitude"], ["magnitude", "confidence"]):
            got, idx = model(pts, items, "standard_vector")
            want = {"confidence": ("confidences", "part_labels"), "direction": ("direction",), "magnitude": ("magnitude",)}
            keys = sorted(k for it in items for k in want[it])
            assert sorted(got.keys()) == keys
            for k in keys:
                assert torch.equal(got[k], full[k]), (items, k)
            assert torch.equal(idx, idx_full)
        default, _ = model(pts)                       # reference default arguments
        assert sorted(default.keys()) == ["direction", "magnitude"]


def test_encoder_equivariance_property(tmp_path):
    """Oracle-free pin: rotating the scan by an icosahedral anchor R_g permutes the encoder features over
    anchors, feats'[..., a] = feats[..., a'] with R_a' = R_g^T R_a, and rotates `direction` by R_g wherever the so3_mean
    projection is well conditioned.  Also validates the intra-neighbour index table.  The discrete sampling chain (FPS:
    hundreds of dependent argmax steps) is only rotation-invariant up to fp32 rounding of the rotated coordinates
    (at N = 5 000 a tie flips), so the property is checked at N = 1 024 on the first seed whose rotated scan
    makes identical sampling decisions."""
    from etch_amd import constants as K
    args, model = make(tmp_path)
    A = K.get_anchors().astype(np.float64)
    g = 17
    Rg = A[g]
    perm = np.array([int(np.argmin(np.abs(A - (Rg.T @ A[a])[None]).reshape(60, -1).sum(1))) for a in range(60)])
    assert sorted(perm.tolist()) == list(range(60))
    checked = False
    for seed in range(1000, 1006):
        x = scan(seed, 1024)
        xr = (x.astype(np.float64) @ Rg.T).astype(np.float32)
        with torch.no_grad():
            e0, _ = model.encoder(torch.from_numpy(x[None]).cuda())
            e1, _ = model.encoder(torch.from_numpy(xr[None]).cuda())
        same_xyz = np.abs(e1.xyz.cpu().numpy()[0].T - e0.xyz.cpu().numpy()[0].T @ Rg.T.astype(np.float32)).max()
        if same_xyz > 1e-5:
            continue                                                   # a sampling tie flipped under rounding: not comparable
        with torch.no_grad():
            r0, _ = model(torch.from_numpy(x[None]).cuda(), ["direction"], "standard_vector")
            r1, _ = model(torch.from_numpy(xr[None]).cuda(), ["direction"], "standard_vector")
        f0, f1 = e0.feats.cpu().numpy(), e1.feats.cpu().numpy()      # [1, 64, 256, 60]
        scale = np.abs(f0).max()
        assert np.abs(f1 - f0[..., perm]).max() / scale < 2e-4
        assert np.abs(f1 - f0).max() / scale > 0.05                   # ... and the permutation is not the identity
        d0, d1 = r0["direction"].cpu().numpy()[0], r1["direction"].cpu().numpy()[0]
        err = np.abs(d1 - d0 @ Rg.T.astype(np.float32)).max(1)
        assert np.median(err) < 1e-3 and (err < 1e-2).mean() > 0.5    # ill-conditioned projections excluded
        checked = True
        break
    assert checked, "no seed with rotation-stable sampling decisions"


def test_degenerate_marker_sets_stay_finite(tmp_path):
    """Scans whose labels hit very few / no markers: masked rows vanish from the normal equations, nothing turns NaN."""
    from etch_amd.models.fit_SMPL import fit_smpl
    args, _ = make(tmp_path)
    B, K_ = 3, 500
    pts = torch.from_numpy(np.stack([scan(300 + b, K_) for b in range(B)])).cuda()
    labels = torch.zeros(B, K_, dtype=torch.int64).cuda()              # scan 0: only marker 0 present
    labels[1] = torch.arange(K_).cuda() % 4                            # scan 1: four markers
    labels[2] = 200                                                    # scan 2: no valid label at all
    conf = torch.rand(B, K_, 1, generator=torch.Generator().manual_seed(0)).cuda() + 0.1
    meshes, markers, valid, info = fit_smpl(args, pts, labels, conf, "neutral")
    assert valid.sum(1).tolist() == [1, 4, 0]
    for a in info:
        assert np.isfinite(a).all()
    assert np.abs(info[0][2]).max() == 0 and np.abs(info[3][2]).max() == 0      # nothing to fit -> parameters stay at zero
    assert all(np.isfinite(m.vertices).all() for m in meshes)


def test_bench_gpus2_self_launch_on_one_gpu():
    """`python bench.py --gpus 2` from a plain interpreter on this 1-GPU box: the parent (which never touches the GPU) starts both
    ranks; ETCH_ALL_RANKS_DEVICE0 puts them on the one GPU and gloo carries the barrier / max / all_gather.  Real hot path, real
    sharding: rank 1 processes scans [2, 4) of every step's batch."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(ETCH_ALL_RANKS_DEVICE0="1", ETCH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--points", "1024"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["launcher"] == "self"
    assert out["gathered_rows"]["scans_reported"] == 4 and out["roofline"]["kernel"]
    assert "cpu_baseline" not in out                      # rank 0 at N = 1 only


def test_evaluate_dataset_on_the_reference_folder_layout(tmp_path):
    """eval.py's loop on a dataset folder in the reference's layout: every activated id evaluated once, per-sample outputs written,
    v2v_score.txt closed by the reference's summary block, V2V consistent with the written body."""
    from test_host_logic import make_eval_tree
    from etch_amd import eval as E
    from etch_amd import inference_demo as D
    from etch_amd.utils.body_model import SyntheticSMPL
    args, model = make(tmp_path / "m")
    args.body_model = {"male": SyntheticSMPL(7), "female": SyntheticSMPL(8)}
    args.output_folder = str(tmp_path / "out")
    scan_dir, smpl_dir, info_dir, ids_pkl = make_eval_tree(tmp_path)
    ds = E.EvalDataset(scan_dir, smpl_dir, info_dir, None, num_point=700, seed=0)
    recs, avg = E.evaluate_dataset(args, model, ds, batch_size=2)
    assert [r["id"] for r in recs] == ["id_a", "id_b", "id_c"]
    assert avg == pytest.approx(np.mean([r["v2v"] for r in recs]))
    lines = open(tmp_path / "out" / "v2v_score.txt").read().splitlines()
    assert lines[3] == "==========" and lines[4].startswith("average v2v: ") and lines[6] == "sample num: 3"
    body = D.load_obj(str(tmp_path / "out" / "id_b" / "forwarded_smpl_mesh_on_pred_id_b.obj"))
    assert abs(E.v2v(ds[1]["gt_vertices"], body.vertices) - recs[1]["v2v"]) < 1e-5


@pytest.mark.parametrize("B,N", [(1, 1024), (2, 1500)])
def test_hip_graph_replay_matches_eager_path(tmp_path, B, N):
    """etch_amd.graph.GraphedHotPath: the whole hot path of a fixed shape captured into one HIP graph and replayed on new inputs returns
    what predict_smpl_batch returns, bit for bit (meshes, markers, fitted parameters), for several different batches."""
    from etch_amd.graph import GraphedHotPath
    from etch_amd.inference_demo import predict_smpl_batch
    args, model = make(tmp_path)
    g = GraphedHotPath(args, model, B, N)
    held = None
    for rep in range(3):
        pts = torch.from_numpy(np.stack([scan(900 + 10 * rep + b, N) for b in range(B)])).cuda()
        meshes, markers, valid, info = g(pts)                   # __call__ returns clones, never the graph's static buffers
        if held is not None:
            # outputs kept from the previous call survive this replay
            assert torch.equal(held[0], held[2]) and torch.equal(held[1], held[3])
        held = (markers, valid, markers.clone(), valid.clone())
        assert markers.data_ptr() != g.dev["markers"].data_ptr() and valid.data_ptr() != g.dev["valid"].data_ptr()
        meshes0, markers0, valid0, info0 = predict_smpl_batch(args, model, pts)
        assert torch.equal(valid, valid0) and torch.equal(markers[valid], markers0[valid0])
        for a, b in zip(info, info0):
            assert np.array_equal(a, b, equal_nan=True)
        for m, m0 in zip(meshes, meshes0):
            assert np.array_equal(m.vertices, m0.vertices, equal_nan=True)


def test_smallest_scans_and_the_too_small_error(tmp_path):
    """256 points is the smallest scan the five-level nets can take (one point left at the deepest level): parity with the oracle there,
    a ValueError -- not a GPU fault -- below."""
    from oracle import stage1 as S1
    args, model = make(tmp_path)
    sd = {k: v.cpu() for k, v in seeded_state_dict(model, 1).items()}
    for B, N in ((1, 256), (2, 257)):
        pts = torch.from_numpy(np.stack([scan(70 + b, N) for b in range(B)]))
        with torch.no_grad():
            res, _ = model(pts.cuda(), ["confidence", "direction", "magnitude"], "standard_vector")
        ref = S1.forward(sd, pts, S1.build_layer_table(), return_aux=True)
        for k in ("part_labels", "confidences", "magnitude"):
            assert float((res[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()) < 1e-4, (N, k)
    with pytest.raises(ValueError, match="at least 256 points"):
        with torch.no_grad():
            model(torch.from_numpy(scan(1, 130)[None]).cuda(), ["confidence"], "standard_vector")
