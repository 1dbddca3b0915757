"""CPU tests of the host-side logic that needs no GPU: layer geometry vs the reference's parameter dump, architecture
constants, the seeded weight generator, the body-model container, OBJ / sampling helpers, eval metrics."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_build_params_reproduces_reference_dump_exactly():
    from etch_amd.models.so3net import build_params
    ref = json.load(open(os.path.join(GOLDEN, "epn_model_setting.json")))
    mine = build_params(mlps=[[32, 32], [64, 64]], strides=[2, 2])
    assert json.dumps(mine) == json.dumps(ref)          # same keys, same order, same floats
    # independent restatement in the oracle agrees for other depths / radii too
    from oracle import stage1 as S
    for radius, mlps, strides, n in ((0.4, [[32, 32], [64, 64], [128, 128]], [2, 2, 2], 1024), (0.55, [[32, 32]], [2], 1024), (0.4, [[32, 32], [64, 64]], [2, 2], 4096)):
        a = build_params(input_radius=radius, input_num=n, mlps=mlps, strides=strides)["backbone"]
        b = S.build_layer_table(radius, mlps, strides, n)
        for blk_a, blk_b in zip(a, b):
            for ca, cb in zip(blk_a, blk_b):
                for k, v in cb.items():
                    assert ca["args"][k] == v, (radius, k)


def test_constants_match_golden(golden):
    from etch_amd import constants as K
    g = golden("constants.npz")
    assert np.array_equal(K.get_anchors(), g["anchors"]) and np.array_equal(K.get_intra_idx(), g["intra_idx"])
    for tag, rad in (("b0c0", 0.08000000000000002), ("b0c1", 0.11313708498984763), ("b1c0", 0.16000000000000003)):
        assert np.array_equal(K.get_kernel_points(rad), g[f"kernels_{tag}"])
    ms = K.default_markerset()
    assert list(ms.keys()) == json.load(open(os.path.join(GOLDEN, "marker_names.json"))) and list(ms.values()) == g["marker_vids"].tolist()


def test_seeded_weights_are_name_keyed_and_deterministic():
    import torch
    from etch_amd.utils.weights import seeded_tensor
    a = seeded_tensor("encoder.x.W", (4, 6), torch.float32, 1)
    assert torch.equal(a, seeded_tensor("encoder.x.W", (4, 6), torch.float32, 1))
    assert not torch.equal(a, seeded_tensor("encoder.y.W", (4, 6), torch.float32, 1))
    assert not torch.equal(a, seeded_tensor("encoder.x.W", (4, 6), torch.float32, 2))
    assert float(a.abs().max()) <= np.sqrt(6.0 / 10) + 1e-6                       # Xavier-uniform bound
    v = seeded_tensor("bn.running_var", (8,), torch.float32, 1)
    assert float(v.min()) >= 0.5


def test_synthetic_body_model_shapes():
    from etch_amd.utils.body_model import SMPL_PARENTS, SyntheticSMPL
    bm = SyntheticSMPL(7)
    assert bm.v_template.shape == (6890, 3) and bm.shapedirs.shape == (6890, 3, 10) and bm.posedirs.shape == (207, 20670)
    assert bm.J_regressor.shape == (24, 6890) and bm.lbs_weights.shape == (6890, 24) and bm.faces.shape[1] == 3
    assert np.allclose(bm.lbs_weights.sum(1), 1, atol=1e-6) and (np.count_nonzero(bm.lbs_weights, axis=1) <= 4).all()
    assert (SMPL_PARENTS[1:] < np.arange(1, 24)).all()                            # parents precede children (LM kernel relies on it)
    assert np.array_equal(SyntheticSMPL(7).v_template, bm.v_template)


def test_obj_roundtrip_and_surface_sampler(tmp_path):
    from etch_amd.inference_demo import load_obj, preprocess_scan, sample_points_from_mesh
    from etch_amd.models.fit_SMPL import Mesh
    v = np.array([[0, 0, 0], [2, 0, 0], [0, 4, 0], [0, 0, 6]], np.float64)
    f = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    p = tmp_path / "t.obj"
    Mesh(v, f).export(str(p))
    m = load_obj(str(p))
    assert np.allclose(m.vertices, v) and np.array_equal(m.faces, f)
    centred, centre = preprocess_scan(str(p))
    assert np.allclose(centre, [1, 2, 3]) and np.allclose(centred.vertices, v - centre)   # bbox mid-point (inference_demo.py:25-28)
    pts = sample_points_from_mesh(m, 4000, seed=3)
    assert pts.shape == (4000, 3) and np.array_equal(pts, sample_points_from_mesh(m, 4000, seed=3))
    assert (pts >= -1e-9).all() and (pts[:, 0] / 2 + pts[:, 1] / 4 + pts[:, 2] / 6 <= 1 + 1e-9).all()   # inside the tetrahedron hull
    # area weighting: the largest face (1,2,3) receives the largest share
    on_big = np.isclose(pts[:, 0] / 2 + pts[:, 1] / 4 + pts[:, 2] / 6, 1.0, atol=1e-9).mean()
    assert on_big > 0.35


def test_eval_metrics():
    from etch_amd.eval import mpjpe, v2v
    assert v2v(np.ones((5, 3)), np.zeros((5, 3))) == np.sqrt(3.0)
    j = np.zeros((45, 3)); k = j.copy(); k[30:] = 9.0                             # joints beyond the first 22 are ignored
    assert mpjpe(j, k) == 0.0


def test_load_smpl_pkl_key_layout(tmp_path):
    """load_smpl_pkl on a synthetic pickle in the SMPL key layout (what SMPL_NEUTRAL_10pc_rmchumpy.pkl holds, fit_SMPL.py:92-101;
    smplx.SMPL.__init__ upstream): posedirs stored (V,3,207) -> (207, V*3) as smplx reshapes it, J_regressor as a scipy sparse
    matrix, kintree_table (2,24) uint32 whose root parent is 2^32 - 1, more than 10 shape components on disk."""
    import pickle

    import scipy.sparse as sp
    import torch

    from etch_amd.utils.body_model import SMPL_PARENTS, SyntheticSMPL, load_smpl_pkl
    from oracle import stage2 as S2
    V = 500
    bm = SyntheticSMPL(3, V=V)
    rng = np.random.default_rng(0)
    extra_shapes = rng.standard_normal((V, 3, 6)).astype(np.float32)                  # the pickle holds more components than used
    kin = np.stack([SMPL_PARENTS.astype(np.int64) % (1 << 32), np.arange(24)]).astype(np.uint32)
    d = {"v_template": bm.v_template.astype(np.float64), "shapedirs": np.concatenate([bm.shapedirs, extra_shapes], 2),
         "posedirs": bm.posedirs.T.reshape(V, 3, 207).copy(), "J_regressor": sp.csc_matrix(bm.J_regressor),
         "weights": bm.lbs_weights, "kintree_table": kin, "f": bm.faces.astype(np.uint32)}
    path = tmp_path / "SMPL_SYNTH.pkl"
    with open(path, "wb") as f:
        pickle.dump(d, f, protocol=2)
    got = load_smpl_pkl(str(path))
    assert got.parents.tolist() == SMPL_PARENTS.tolist() and got.parents.dtype == np.int32
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
        assert getattr(got, k).dtype == np.float32 and np.array_equal(getattr(got, k), getattr(bm, k)), k
    assert got.posedirs.shape == (207, V * 3) and got.shapedirs.shape == (V, 3, 10) and got.num_betas == 10
    assert np.array_equal(got.faces, bm.faces)
    # the element order is the one LBS consumes: posed vertices of the loaded model equal those of the source model
    x = torch.from_numpy(rng.standard_normal((2, 85)).astype(np.float32) * 0.3)
    a = S2.lbs(S2.TorchBody(got), x[:, 69:79], torch.cat([x[:, 79:82], x[:, :69]], 1), x[:, 82:85])[0]
    b = S2.lbs(S2.TorchBody(bm), x[:, 69:79], torch.cat([x[:, 79:82], x[:, :69]], 1), x[:, 82:85])[0]
    assert torch.equal(a, b)


def test_reference_operator_names_are_exported():
    """every operator name of the reference that sits on or beside the path resolves under the same module path
    (importing needs no GPU; calling them does)."""
    import importlib
    want = {
        "etch_amd.vgtk_so3conv": ["SphericalPointCloud", "BasicSO3Conv", "InterSO3Conv", "IntraSO3Conv", "functional", "batch_gather", "group_nd",
                                  "ball_query_index", "furthest_sample_index", "furthest_sample", "get_anchors", "get_intra_idx"],
        "etch_amd.vgtk_so3conv.functional".replace(".functional", ""): [],
        "etch_amd.vgtk_functional": ["batched_index_select", "inter_so3conv_feat_grouping", "get_occupancy_features", "add_shadow_point",
                                     "add_shadow_feature", "get_sphereical_kernel_points_from_ply", "ball_query", "inter_spconv_grouping_ball",
                                     "inter_so3conv_grouping", "inter_so3conv_grouping_anchor", "intra_so3conv_grouping", "get_anchors",
                                     "get_intra_idx"],
        "etch_amd.models.so3conv": ["preprocess_input", "IntraSO3ConvBlock", "InterSO3ConvBlock", "BasicSO3ConvBlock", "SeparableSO3ConvBlock",
                                    "so3_mean"],
        "etch_amd.models.pointnet2_utils": ["square_distance", "index_points", "PointFeatPropagation"],
        "etch_amd.models.pointops": ["furthestsampling", "knnquery", "queryandgroup", "interpolation"],
        "etch_amd.models.direction_backbones": ["BatchLinear", "BatchMLP", "DotProdAttention", "MultiHeadAttention", "StackedMHSA"],
        "etch_amd.models.fit_SMPL": ["get_markers", "fit_smpl"],
        "etch_amd.models.so3net": ["build_model", "EquivBackbone"],
        "etch_amd.epn_grouping": ["ball_query", "furthest_point_sampling"],
        "etch_amd.epn_gathering": ["gather_points_forward", "gather_points_backward"],
        "etch_amd.pointops_cuda": ["knnquery_cuda", "furthestsampling_cuda"],
    }
    for mod, names in want.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)
    import etch_amd.vgtk_so3conv as V
    assert V.functional.get_anchors(60).shape == (60, 3, 3) and V.functional.get_intra_idx().shape == (60, 12)
    kp = V.functional.get_sphereical_kernel_points_from_ply(0.7 * 0.2, 1)
    assert kp.shape == (24, 3) and abs(float(np.sqrt((kp ** 2).sum(1).max())) - 0.14) < 1e-6


def test_every_encoder_depth_constructs_with_the_reference_state_dict(tmp_path):
    """The reference builds EPN_layer_num 1..4 (models_pointcloud.py:34-48: feature widths 32 / 64 / 128 / 256); so does this build: same
    state-dict keys / shapes / dtypes as the reference's constructor emitted (tests/golden/state_dict_manifest*.json) and the same
    EPN_model_setting_json dump per depth.  Anything else is rejected at construction; the CLI takes 1-4."""
    import types

    import pytest

    from etch_amd import constants as K
    from etch_amd import inference_demo as D
    from etch_amd.models.models_pointcloud import GT_network_equiv
    for n in (1, 2, 3, 4):
        out = tmp_path / f"l{n}"
        opt = types.SimpleNamespace(output_folder=str(out), EPN_input_radius=0.4, EPN_layer_num=n, device="cpu", markerset=K.default_markerset())
        m = GT_network_equiv(option=opt)
        mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
        tag = "" if n == 2 else f"_l{n}"
        assert mine == json.load(open(os.path.join(GOLDEN, f"state_dict_manifest{tag}.json"))), n
        dump = json.load(open(out / "EPN_model_setting_json"))
        assert json.dumps(dump) == json.dumps(json.load(open(os.path.join(GOLDEN, f"epn_model_setting{tag}.json")))), n
    for n in (0, 5):
        opt = types.SimpleNamespace(output_folder=str(tmp_path), EPN_input_radius=0.4, EPN_layer_num=n, device="cpu", markerset=K.default_markerset())
        with pytest.raises(ValueError, match="EPN_layer_num"):
            GT_network_equiv(option=opt)
    with pytest.raises(SystemExit):
        D.main(["--scan_path", "x.obj", "--EPN_layer_num", "5"])


def _write_obj(path, v, f):
    with open(path, "w") as fh:
        for p in v:
            fh.write(f"v {p[0]} {p[1]} {p[2]}\n")
        for t in f:
            fh.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")


def make_eval_tree(root, ids=("id_b", "id_a", "id_c"), V=6890):
    """A dataset folder in the reference's layout (README.md:74-90) with synthetic content."""
    import pickle
    rng = np.random.default_rng(0)
    scan_dir, smpl_dir, info_dir = root / "ds" / "model", root / "ds" / "smplh", root / "gt" / "npz"
    info_dir.mkdir(parents=True)
    for k, i in enumerate(ids):
        (scan_dir / i).mkdir(parents=True)
        (smpl_dir / i).mkdir(parents=True)
        v = (rng.standard_normal((400, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)
        _write_obj(scan_dir / i / f"{i}.obj", v, np.stack([np.arange(0, 398), np.arange(1, 399), np.arange(2, 400)], 1))
        sv = rng.standard_normal((V, 3)).astype(np.float32)
        _write_obj(smpl_dir / i / f"mesh_smpl_{i}.obj", sv, [[0, 1, 2]])
        np.savez(smpl_dir / i / f"info_{i}.npz", gender=np.array([k % 2], np.int32), joints=rng.standard_normal((73, 3)).astype(np.float32),
                 betas=np.zeros(10, np.float32))
        np.savez(info_dir / f"{i}.npz", info_points=np.zeros((4, 3)), info_vectors=np.zeros((4, 3)))
    (scan_dir / "not_in_smpl").mkdir()                     # no body model -> skipped (GT_dataloader_mixed.py:120-127)
    ids_pkl = root / "val_ids.pkl"
    with open(ids_pkl, "wb") as f:
        pickle.dump(list(ids[:2]) + ["absent"], f)         # id_c is not activated
    return str(scan_dir), str(smpl_dir), str(info_dir), str(ids_pkl)


def test_eval_dataset_reads_the_reference_layout(tmp_path):
    from etch_amd.eval import EvalDataset
    scan_dir, smpl_dir, info_dir, ids_pkl = make_eval_tree(tmp_path)
    ds = EvalDataset(scan_dir, smpl_dir, info_dir, ids_pkl, num_point=300, seed=3)
    assert ds.id_list == ["id_a", "id_b"]                  # sorted, activated, complete entries only
    a, b = ds[0], ds[1]
    assert a["id"] == "id_a" and a["gender"] == "male" and b["gender"] == "female"     # gender 1 -> male (:133)
    assert a["hitpts"].shape == (300, 3) and a["hitpts"].dtype == np.float32 and a["gt_vertices"].shape == (6890, 3) and a["gt_joints"].shape == (73, 3)
    assert np.array_equal(ds[0]["hitpts"], a["hitpts"])    # seeded: reproducible
    assert len(EvalDataset(scan_dir, smpl_dir, info_dir)) == 3


def test_transition_down_offsets_and_too_small_scans():
    """TransitionDown's per-scan counts (pointtransformer_seg.py:55-59) and the loud failure for scans that run out of points before
    the last level (the reference is left with empty segments there; a GPU path must not read out of bounds instead)."""
    import pytest
    from etch_amd.models.pointtransformer_seg import downsampled_offsets
    assert downsampled_offsets([5000, 10000, 15001], 4) == [1250, 2500, 3750]
    assert downsampled_offsets([7, 19], 4) == [1, 4]
    with pytest.raises(ValueError, match="at least 256 points"):
        downsampled_offsets([1250, 1253], 4)


def test_load_smplx_reads_the_distributed_file_layout(tmp_path):
    """load_smplx on files written in the layout of SMPLX_NEUTRAL.npz / .pkl (400 shape + expression components, 486 pose-basis columns,
    55-joint kintree): the coefficient vector is [10 betas | 10 expression], arrays land in the smplx buffer layout, a wrong tree is refused."""
    import pickle

    import pytest

    from etch_amd.utils import body_model as BM
    rng = np.random.default_rng(0)
    V = 10475
    d = dict(v_template=rng.standard_normal((V, 3)), shapedirs=rng.standard_normal((V, 3, 400)).astype(np.float32),
             posedirs=rng.standard_normal((V, 3, 486)).astype(np.float32), J_regressor=rng.random((55, V)).astype(np.float32),
             weights=rng.random((V, 55)).astype(np.float32), kintree_table=np.stack([np.where(BM.SMPLX_PARENTS < 0, 2 ** 32 - 1, BM.SMPLX_PARENTS),
                                                                                     np.arange(55)]).astype(np.uint32),
             f=rng.integers(0, V, (20908, 3)).astype(np.uint32))
    np.savez(tmp_path / "SMPLX_NEUTRAL.npz", **d)
    with open(tmp_path / "SMPLX_NEUTRAL.pkl", "wb") as fh:
        pickle.dump(d, fh)
    for name in ("SMPLX_NEUTRAL.npz", "SMPLX_NEUTRAL.pkl"):
        bm = BM.load_smplx(str(tmp_path / name))
        assert (bm.num_joints, bm.num_betas, bm.num_verts) == (55, 20, V)
        assert np.array_equal(bm.parents, BM.SMPLX_PARENTS)
        assert np.array_equal(bm.shapedirs[:, :, :10], d["shapedirs"][:, :, :10]) and np.array_equal(bm.shapedirs[:, :, 10:], d["shapedirs"][:, :, 300:310])
        assert bm.posedirs.shape == (486, V * 3) and bm.posedirs[7, 3 * 5 + 2] == d["posedirs"][5, 2, 7]
        assert np.array_equal(bm.extra_vids, BM.SMPLX_EXTRA_JOINT_VIDS) and bm.extra_vids.max() < V
    d["kintree_table"][0, 30] = 3
    np.savez(tmp_path / "bad.npz", **d)
    with pytest.raises(ValueError, match="kinematic tree"):
        BM.load_smplx(str(tmp_path / "bad.npz"))


def _bf16_to_f64(p):
    import torch
    return (p.to(torch.int32) << 16).view(torch.float32).double()


def test_split3_bf16_is_exact():
    """ops.split3_bf16: x = hi + mid + lo exactly, every part a bf16 bit pattern (the operand split of the *_split kernels; the device-side
    twins are split3_pack8 / ws_split3_pack4 / ml_split / fd_split in csrc/)."""
    import torch
    from etch_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(20000, generator=g) * torch.exp(torch.randn(20000, generator=g) * 6)
    x = torch.cat([x, torch.tensor([0.0, -0.0, 1.0, -1.0, 1e-30, -3e38, 2.0 ** -126, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24])])
    pl = ops.split3_bf16(x)
    assert pl.dtype == torch.int16 and tuple(pl.shape) == (3,) + tuple(x.shape)
    assert torch.equal(sum(_bf16_to_f64(p) for p in pl), x.double())
    hi, mid, lo = (_bf16_to_f64(p).abs() for p in pl)
    nz = x != 0
    assert bool((mid[nz] <= hi[nz] * 2.0 ** -7).all()) and bool((lo[nz] <= hi[nz] * 2.0 ** -15).all())      # 8 bits each


def test_split_weight_layouts_address_the_right_elements():
    """inter_weight_split / intra_weight_split / lrd_weight_split: element (fragment indices) of the device layout = the plane of the weight
    the kernels' index formulas name (csrc/so3conv.hip BX step 2, csrc/so3conv_ws.hip, csrc/fused_dense.hip)."""
    import torch
    from etch_amd import ops
    g = torch.Generator().manual_seed(1)
    rng = np.random.default_rng(0)
    # inter: [tg][mt][pl][lane = 16 kg + ol][8] of W[:, cols] with the contraction order of inter_weight_frag
    for cin, cout in ((32, 32), (64, 64), (128, 32)):
        W = torch.randn(cout, cin * 24, generator=g)
        q = ops.inter_weight_split(W, cin).view(cin * 24 // 32, cout // 16, 3, 64, 8)
        Wk = ops.split3_bf16(W[:, ops._inter_contraction_cols(cin, 24, W.device)])
        for _ in range(50):
            tg, mt, pl, lane, e = (int(rng.integers(n)) for n in q.shape)
            assert q[tg, mt, pl, lane, e] == Wk[pl, 16 * mt + lane % 16, 32 * tg + 8 * (lane // 16) + e]
    # intra: [mt][kq][ks][pl][lane = 32 kg + i][8] of W2[32 mt + i][3 kq C + 16 ks + 8 kg + e]
    for C in (32, 64):
        W2 = torch.randn(C, 12 * C, generator=g)
        nks = 3 * C // 16
        q = ops.intra_weight_split(W2).view(C // 32, 4, nks, 3, 64, 8)
        P = ops.split3_bf16(W2)
        for _ in range(50):
            mt, kq, ks, pl, lane, e = (int(rng.integers(n)) for n in q.shape)
            assert q[mt, kq, ks, pl, lane, e] == P[pl, 32 * mt + lane % 32, 3 * kq * C + 16 * ks + 8 * (lane // 32) + e]
    # linear_relu_dot: [g][t][strip][pl][lane = 16 kg + col][8] of w[g J + 16 strip + col][32 t + 8 kg + e]
    for G, K in ((3, 64), (2, 128)):
        w = torch.randn(G * 128, K, generator=g)
        q = ops.lrd_weight_split(w).view(G, K // 32, 8, 3, 64, 8)
        P = ops.split3_bf16(w)
        for _ in range(50):
            gi, t, strip, pl, lane, e = (int(rng.integers(n)) for n in q.shape)
            assert q[gi, t, strip, pl, lane, e] == P[pl, gi * 128 + 16 * strip + lane % 16, 32 * t + 8 * (lane // 16) + e]


def test_row_powers_of_two_are_capped_like_the_device_side():
    """ops._row_pow2 / _pow2_exp mirror the device's etch_scale_exp (csrc/split_bf16.h) -- rows whose maximum is zero, subnormal or
    non-finite keep the factor 1, the exponent is capped at 120 (dirtail: 60, its power is folded into a bias), every scaled row and every
    inverse power is finite and the scaling is exact."""
    import torch

    from etch_amd import ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn(8, 64, generator=g)
    w[0] *= 1e-38
    w[1] = 1e-45
    w[2] *= 1e30
    w[3] = 0
    w[4] *= 3e-37
    w[5, 0] = float("inf")
    ws, wsc = ops._row_pow2(w)
    assert bool(torch.isfinite(wsc).all()) and bool(torch.isfinite(ws[[0, 1, 2, 3, 4, 6, 7]]).all())
    assert torch.equal(wsc[[1, 3, 5]], torch.ones(3))                                   # subnormal / zero / non-finite rows: factor 1
    assert bool((torch.log2(wsc) >= -120).all()) and bool((torch.log2(wsc) == torch.log2(wsc).round()).all())
    m = ws[[2, 6, 7]].abs().amax(1)
    assert bool(((m >= 8) & (m < 16)).all())
    keep = [0, 1, 2, 3, 4, 6, 7]
    assert torch.equal((ws * wsc[:, None])[keep], w[keep])                               # exact
    _, wsc60 = ops._row_pow2(w, cap=60)
    assert bool((torch.log2(wsc60) >= -60).all())
    Wf = torch.randn(128, 64, generator=g)
    Wf[7] *= 1e-37
    q = ops.dirtail_weight_split(Wf)
    tab = ops.dirtail_constants(torch.randn(128, generator=g) * 100, torch.randn(128, generator=g), torch.tensor(0.5), q.wsc)
    assert bool(torch.isfinite(tab).all())
