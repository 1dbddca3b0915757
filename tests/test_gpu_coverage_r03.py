"""Round-3 coverage additions under -m gpu ("Next round" 1a-1d):

  * configs[4] at its OWN size: the 8 x 20 000 shard with the 75+125-iteration 188-DoF fit through predict_smpl_batch and
    HotPathPipeline (bitwise pipeline == synchronous, scan 0 alone == scan 0 in the batch, stage 1 of scan 0 vs the reference's
    Python, the fit vs the oracle over a prefix that crosses the stage hand-over);
  * checkpoint-supplied `intra_idx` / `anchors` / `kernels` buffers are honoured end to end (GPU == oracle with the same buffers);
  * RCCL executed once on this 1-GPU box: bench.py's collective path on a world-size-1 `nccl` process group.
The stage-2 conditioning stress fixture lives in test_gpu_stage2.py."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from _parity import check_stage1_vs_fixture, oracle_trace
from etch_amd.utils.weights import load_seeded, seeded_state_dict

pytestmark = pytest.mark.gpu
ITEMS = ["confidence", "direction", "magnitude"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(seed, n, sigma=(0.14, 0.31, 0.085)):
    return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array(sigma)).astype(np.float32)


def make(tmp_path, seed=1, body=None, layers=2):
    from etch_amd import constants as K
    from etch_amd.models.models_pointcloud import GT_network_equiv
    from etch_amd.utils.body_model import SyntheticSMPL
    args = types.SimpleNamespace(output_folder=str(tmp_path), EPN_input_radius=0.4, EPN_layer_num=layers, device=torch.device("cuda"),
                                 markerset=K.default_markerset(), scale_magnitude=10, body_model=body or SyntheticSMPL(7))
    return args, load_seeded(GT_network_equiv(option=args), seed).cuda().eval()


def _same(a, b):
    f = lambda t: np.asarray(t.cpu() if torch.is_tensor(t) else t)
    return np.array_equal(f(a), f(b), equal_nan=True)


def test_config4_full_size_shard_8x20000_with_188dof_fit(tmp_path, golden):
    """BASELINE configs[4]'s per-GPU shard exactly as `bench.py --config 4` runs it: 8 dense 20 000-point scans (bench seeds 1000 + b),
    SMPL-X-sized 188-DoF body, 75 + 125 LM iterations."""
    from etch_amd.inference_demo import predict_smpl_batch
    from etch_amd.pipeline import HotPathPipeline
    from etch_amd.utils.body_model import SyntheticSMPLX
    from oracle import stage2 as S2
    g, c = golden("model_n20000.npz"), golden("constants.npz")
    args, model = make(tmp_path, int(g["seed"]), body=SyntheticSMPLX(7))
    B, N, IT = 8, 20000, (75, 125)
    pts = np.stack([scan(1000 + b, N) for b in range(B)])
    assert np.array_equal(pts[:1], g["points"])
    dev = torch.from_numpy(pts).cuda()
    # stage 1 of scan 0 against the reference's own Python; a scan's result must not depend on its batch neighbours
    with torch.no_grad():
        res, _ = model(dev, ITEMS, "standard_vector")
        anc_w = model.last_anc_w.clone()
        print("config4 shard, scan 0 vs reference:", check_stage1_vs_fixture(res, anc_w, g, c["anchors"], scans=[0]))
        solo, _ = model(dev[:1].contiguous(), ITEMS, "standard_vector")
    for k in res:
        assert torch.equal(solo[k][0], res[k][0]), k
    # the full schedule, synchronous
    kw = dict(steps_stage0=IT[0], steps_stage1=IT[1])
    meshes, markers, valid, info, aux = predict_smpl_batch(args, model, dev, "neutral", return_trace=True, **kw)
    assert len(meshes) == B and meshes[0].vertices.shape == (10475, 3) and markers.shape == (B, 86, 3)
    assert [a.shape for a in info] == [(B, 54, 3), (B, 20), (B, 3), (B, 3), (B, 76, 3)]
    assert aux["err_trace"].shape == (B, IT[0] + IT[1] + 2)
    assert aux["status"].cpu().tolist() == [0] * B
    assert all(np.isfinite(a).all() for a in info)
    # the 2-deep stream pipeline on three batches of this size returns the same bits
    other = torch.from_numpy(np.stack([scan(1000 + B + b, N) for b in range(B)])).cuda()
    pipe = HotPathPipeline(args, model, "neutral", max_in_flight=2, **kw)
    got = list(pipe.run(iter([dev, other, dev])))
    for r in (got[0], got[2]):
        assert _same(r[1], markers) and _same(r[2], valid)
        for a, b in zip(r[3], info):
            assert _same(a, b)
        for m0, m1 in zip(r[0], meshes):
            assert _same(m0.vertices, m1.vertices)
    # the fit of scans 0 and 7 against the oracle on the GPU's own markers, over a prefix that ends AFTER stage 1 has started
    # (the oracle differentiates the full 10 475-vertex mesh: ~0.4 s per iteration and scan)
    ids, pre = [0, B - 1], (IT[0], 10)
    mk, va = markers[ids].cpu(), valid[ids].cpu()
    assert bool(torch.isfinite(mk).all())
    _, _, _, info_p, aux_p = predict_smpl_batch(args, model, dev, "neutral", return_trace=True, steps_stage0=pre[0], steps_stage1=pre[1])
    mv = np.array(list(args.markerset.values()))
    trace = []
    ref = S2.fit_smpl(args.body_model, mv, mk, va, steps_stage0=pre[0], steps_stage1=pre[1], trace=trace)
    rt, gt = oracle_trace(trace, *pre), aux_p["err_trace"].cpu().numpy()[ids]
    assert gt.shape == rt.shape == (2, pre[0] + pre[1] + 2)
    assert np.abs(gt - rt).max() / rt.max() < 1e-4
    assert np.abs(aux_p["verts"].cpu().numpy()[ids] - ref["verts"].numpy()).max() < 1e-4
    assert np.abs(info_p[4][ids] - ref["joints"].numpy()).max() < 1e-4
    xr = torch.cat([ref["pose"], ref["betas"], ref["orient"], ref["transl"]], 1).numpy()
    dev_x = np.abs(aux_p["x"].cpu().numpy()[ids] - xr)
    print("config4 shard fit (75+10) parameter deviation vs oracle:", {"pose": float(dev_x[:, :162].max()), "betas": float(dev_x[:, 162:182].max()),
                                                                      "orient": float(dev_x[:, 182:185].max()), "transl": float(dev_x[:, 185:].max())})
    assert dev_x.max() < 1e-4
    # the stage-0 result the hand-over starts from, and the full schedule's trace continues the prefix's (same iterations, same bits)
    assert np.abs(aux_p["x_stage0"].cpu().numpy()[ids][:, :162] - ref["x_stage0"].numpy()[:, :162]).max() < 1e-4
    full = aux["err_trace"].cpu().numpy()[ids]
    assert np.array_equal(full[:, :pre[0] + 1 + pre[1] + 1], gt)


def test_well_posed_188dof_fit_across_the_stage_handover_vs_oracle(golden):
    """The 188-DoF fit on WELL-POSED markers (86 minus a few masked, 2 mm noise) over 75 + 10 iterations -- every stage-0 iteration of
    configs[4]'s schedule and the hand-over to the 20-coefficient stage (r02 compared 8 + 10) -- against the committed oracle runs of
    tests/golden/fit_smplx_handover.npz (oracle/gen_fit_smplx_fixture.py; the oracle differentiates the full 10 475-vertex mesh: 2.5 min
    of CPU, hence a fixture).  30 finger joints are seen by few markers: the oracle's own fp32 and fp64 runs differ by 3.5e-4 in those
    pose parameters, so raw parameters are held to the fp64-yardstick rule (as close to the fp64 run as the fp32 oracle, x2); error trace,
    vertices and joints (well conditioned) to 1e-4."""
    from etch_amd import constants as K
    from etch_amd import ops
    from etch_amd.models.fit_SMPL import _device_body
    from etch_amd.utils.body_model import SyntheticSMPLX
    g = golden("fit_smplx_handover.npz")
    it0, it1 = (int(v) for v in g["iters"])
    bm = SyntheticSMPLX(7)
    mv = np.array(list(K.default_markerset().values()))
    db = _device_body(bm, mv, torch.device("cuda"))
    x, x0, tr = ops.smpl_lm_fit(db.lm_consts, torch.from_numpy(g["markers"]).cuda(), torch.from_numpy(g["valid"].astype(np.float32)).cuda(),
                                it0, 0.5, 0.01, it1, 0.2, 1e-3, True, nj=db.nj, nb=db.nb)
    verts, joints = ops.smpl_lbs(db.lbs_consts, x, db.V, db.n_extra, nj=db.nj, nb=db.nb)
    t64 = g["trace_fp64"]
    assert tr.shape == t64.shape == (2, it0 + it1 + 2)
    assert np.abs(tr.cpu().numpy() - t64).max() / t64.max() < 1e-4
    assert np.abs(tr.cpu().numpy() - g["trace_fp32"]).max() / t64.max() < 1e-4
    assert t64[:, it0].max() < 0.05 * t64[:, 0].min()                              # stage 0 did descend before the hand-over
    x, x64, x32 = x.cpu().numpy().astype(np.float64), g["x_fp64"], g["x_fp32"].astype(np.float64)
    report = {}
    for name, sl in (("pose", slice(0, 162)), ("betas", slice(162, 182)), ("orient", slice(182, 185)), ("transl", slice(185, 188))):
        gpu, ref = float(np.abs(x[:, sl] - x64[:, sl]).max()), float(np.abs(x32[:, sl] - x64[:, sl]).max())
        report[name] = "%.1e / %.1e" % (gpu, ref)
        assert gpu <= max(2.0 * ref, 1e-4), (name, gpu, ref)
    print("188-DoF 75+10: |x - x_fp64| per group (gpu / oracle fp32):", report)
    v64 = g["verts_fp64"]
    assert np.abs(verts.cpu().numpy()[:, ::10] - v64).max() < 1e-4
    assert np.abs(joints.cpu().numpy() - g["joints_fp64"]).max() < 1e-4
    d0 = np.abs(x0.cpu().numpy()[:, :162] - g["x_stage0_fp64"][:, :162]).max()
    assert d0 <= max(2.0 * np.abs(g["x_stage0_fp32"] - g["x_stage0_fp64"])[:, :162].max(), 1e-4), d0


def test_checkpoint_supplied_anchor_buffers_are_honoured(tmp_path, golden):
    """A checkpoint whose `intra_idx` columns are permuted (the matching W columns permuted with them -> the same function), and one whose
    `intra_idx` / `kernels` / `anchors` buffers are genuinely different (another column order WITHOUT touching W, kernel points rescaled,
    anchors rotated by a fixed rotation): the GPU must follow the buffers of the state dict exactly as the oracle does with the same
    buffers -- what protects real-checkpoint users from the re-implemented trimesh.face_adjacency column order."""
    from oracle import stage1 as S1
    args, model = make(tmp_path)
    N = 1024
    pts = torch.from_numpy(np.stack([scan(40 + b, N) for b in range(2)]))
    base = {k: v.clone() for k, v in seeded_state_dict(model, 1).items()}
    table = S1.build_layer_table()
    rng = np.random.default_rng(5)

    def run_gpu(sd):
        model.load_state_dict(sd)
        with torch.no_grad():
            res, _ = model(pts.cuda(), ITEMS, "standard_vector")
        return {k: v.cpu() for k, v in res.items()}, model.last_anc_w.cpu().clone()

    def compare(sd, tag):
        res, aw = run_gpu(sd)
        ref = S1.forward({k: v.cpu() for k, v in sd.items()}, pts, table, return_aux=True)
        for k in ("part_labels", "confidences", "magnitude"):
            e = float((res[k] - ref[k]).abs().max() / ref[k].abs().max())
            assert e < 1e-4, (tag, k, e)
        # anchor weights: 1e-4 on (nearly) every point; a handful of points carry saturated attention rows that amplify fp32 rounding -- the
        # reference's OWN fp32 run sits up to 3.8e-4 from its fp64 run on such points (tests/golden/model_n5000.npz, padding_heavy_fp64.npz)
        per_point = (aw - ref["anc_w"]).abs().amax(-1).flatten() / ref["anc_w"].abs().max()
        assert float((per_point < 1e-4).float().mean()) >= 0.999 and float(per_point.max()) < 4e-4, (tag, "anc_w", float(per_point.max()))
        return res, aw

    res0, aw0 = compare(base, "seeded")
    # (1) column permutation of intra_idx with the matching permutation of W's tap axis: the same function, different buffers
    sd1 = {k: v.clone() for k, v in base.items()}
    for k in [k for k in base if k.endswith("intra_conv.conv.intra_idx")]:
        perm = rng.permutation(12)
        sd1[k] = base[k][:, perm].contiguous()
        wk = k.replace("intra_idx", "basic_conv.W")
        W = base[wk]
        co = W.shape[0]
        sd1[wk] = W.view(co, -1, 12)[:, :, perm].reshape(co, -1).contiguous()
    res1, aw1 = compare(sd1, "permuted idx + W")
    for k in ("part_labels", "magnitude"):
        assert float((res1[k] - res0[k]).abs().max() / res0[k].abs().max()) < 1e-4, k          # same function as the unpermuted checkpoint
    # (2) genuinely different buffers: idx columns permuted WITHOUT W, kernel points rescaled, anchors re-ordered
    sd2 = {k: v.clone() for k, v in base.items()}
    aperm = torch.from_numpy(rng.permutation(60))
    for k in base:
        if k.endswith("intra_conv.conv.intra_idx"):
            sd2[k] = base[k][:, rng.permutation(12)].contiguous()
        elif k.endswith("inter_conv.conv.kernels"):
            sd2[k] = (base[k] * 0.8).contiguous()
        elif k.endswith("inter_conv.conv.anchors"):
            sd2[k] = base[k][aperm].contiguous()
        elif k.endswith("intra_conv.conv.anchors"):
            sd2[k] = base[k].flip(0).contiguous()         # the last block's copy is what the direction head's so3_mean reads (models_pointcloud.py:162)
    res2, aw2 = compare(sd2, "different buffers")
    from _parity import direction_within_conditioning
    ref2 = S1.forward({k: v.cpu() for k, v in sd2.items()}, pts, table, return_aux=True)
    last_anchors = sd2["encoder.backbone.1.blocks.1.intra_conv.conv.anchors"].cpu().numpy()
    direction_within_conditioning(res2["direction"].numpy().reshape(-1, 3), aw2.numpy().reshape(-1, 60), ref2["anc_w"].numpy().reshape(-1, 60),
                                  ref2["direction"].numpy().reshape(-1, 3), last_anchors)
    assert float((res2["part_labels"] - res0["part_labels"]).abs().max() / res0["part_labels"].abs().max()) > 1e-3   # ... and they do change the function


def test_rccl_world_size_1_collective_path():
    """RCCL executes once on this box: bench.py with ETCH_FORCE_DIST=1 builds a world-size-1 `nccl` process group and runs its barrier,
    the fp64 all_reduce(MAX) and the device-tensor all_gather of the result rows through it."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ETCH_DIST_BACKEND")}
    env.update(ETCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--points", "1024",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["config"]["collective_backend"].startswith("nccl")
    assert out["n_gpus"] == 1 and out["gathered_rows"]["scans_reported"] == 2 and out["value"] > 0
    # and the three collectives directly, on device tensors, in a fresh process
    code = (
        "import os, torch, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from etch_amd import parallel as P\n"
        "import torch.distributed as dist\n"
        "torch.cuda.set_device(0)\n"
        "P.init()\n"
        "assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1\n"
        "P.barrier()\n"
        "rows = torch.arange(12, dtype=torch.float32, device='cuda').view(4, 3)\n"
        "g = P.gather_rows(rows)\n"
        "assert g.is_cuda and torch.equal(g, rows)\n"
        "assert P.max_over_ranks(1.25, torch.device('cuda', 0)) == 1.25\n"
        "dist.destroy_process_group()\n"
        "print('rccl ok')\n")
    env["MASTER_PORT"] = "29548"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.parametrize("layers,n", [(1, 1024), (3, 1024), (4, 512)])
def test_encoder_depths_1_3_4_vs_reference(tmp_path, golden, layers, n):
    """EPN_layer_num 1 / 3 / 4 (models_pointcloud.py:34-48: feature widths 32 / 128 / 256; 8-head attention over 32 / 128 / 256 dims; conv
    channel pairs up to (256, 256)) against the reference's own Python run of that depth (tests/golden/model_l<d>_n<n>.npz), same 1e-4 bar
    and fp64 conditioning yardstick as the released depth; then the whole pipeline on that model."""
    from etch_amd.inference_demo import predict_smpl_batch
    g, c = golden(f"model_l{layers}_n{n}.npz"), golden("constants.npz")
    args, model = make(tmp_path, int(g["seed"]), layers=layers)
    manifest = json.load(open(os.path.join(ROOT, "tests", "golden", f"state_dict_manifest_l{layers}.json")))
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()] == manifest
    pts = torch.from_numpy(g["points"]).cuda()
    with torch.no_grad():
        res, sel = model(pts, ITEMS, "standard_vector")
        anc_w = model.last_anc_w.clone()
        print(f"depth {layers} deviations:", check_stage1_vs_fixture(res, anc_w, g, c["anchors"]))
        solo, _ = model(pts[:1].contiguous(), ITEMS, "standard_vector")
    for k in res:
        assert torch.equal(solo[k][0], res[k][0]), k
    meshes, markers, valid, info = predict_smpl_batch(args, model, pts, "neutral", steps_stage0=5, steps_stage1=5)
    assert len(meshes) == pts.shape[0] and markers.shape == (pts.shape[0], 86, 3)
