"""Pin the oracle's stage-1 restatement (oracle/stage1.py) against golden vectors emitted by the
reference's own Python (oracle/ref_harness/gen_golden.py).  CPU only; tolerances are tight because
both sides run the same torch-CPU kernels."""
import json
import os

import numpy as np
import pytest
import torch

from etch_amd.utils.weights import seeded_tensor
from oracle import stage1 as S

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _manifest():
    return json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))


def model_state_dict(seed, consts):
    """Rebuild the full state dict from the manifest + seed + architecture constants."""
    sd = {}
    for name, shape, dt in _manifest():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "anchors":
            sd[name] = torch.from_numpy(consts["anchors"])
        elif leaf == "intra_idx":
            sd[name] = torch.from_numpy(consts["intra_idx"])
        elif leaf == "kernels":
            b, c = name.split(".")[2], name.split(".")[4]
            sd[name] = torch.from_numpy(consts[f"kernels_b{b}c{c}"])
        elif leaf == "num_batches_tracked":
            sd[name] = torch.zeros((), dtype=torch.int64)
        else:
            sd[name] = seeded_tensor(name, shape, getattr(torch, dt), seed)
    return sd


def test_layer_table_matches_reference_dump():
    ref = json.load(open(os.path.join(GOLDEN, "epn_model_setting.json")))
    mine = S.build_layer_table()
    for bi, block in enumerate(ref["backbone"]):
        for ci, conv in enumerate(block):
            a = conv["args"]
            for k in ("dim_in", "dim_out", "stride", "radius", "sigma", "n_neighbor", "lazy_sample"):
                assert mine[bi][ci][k] == a[k], (bi, ci, k)


def test_constants_group_properties(golden):
    c = golden("constants.npz")
    A, I = c["anchors"], c["intra_idx"]
    assert A.shape == (60, 3, 3) and A.dtype == np.float32 and I.shape == (60, 12)
    assert np.abs(A @ A.transpose(0, 2, 1) - np.eye(3)).max() < 1e-6
    assert np.allclose(np.linalg.det(A), 1, atol=1e-6)
    assert np.abs(A[29] - np.eye(3)).max() < 1e-6
    assert all(sorted(I[:, k]) == list(range(60)) for k in range(12))
    assert I[0].tolist() == [10, 33, 13, 15, 56, 30, 59, 8, 44, 0, 45, 26]       # SURVEY appendix D fingerprint
    assert c["marker_vids"].shape == (86,) and len(set(c["marker_vids"].tolist())) == 86
    # kernel points: 24 points, radius-normalised to 0.7 * conv radius (functional.py:146-157, modules.py:13,99)
    kp = c["kp24_raw"]
    r = np.sqrt((kp ** 2).sum(1).max())
    for tag, rad in (("b0c0", 0.08000000000000002), ("b0c1", 0.11313708498984763), ("b1c0", 0.16000000000000003), ("b1c1", 0.16000000000000003)):
        assert np.array_equal(c[f"kernels_{tag}"], (kp * (0.7 * rad) / r).astype(np.float32)), tag


@pytest.mark.parametrize("tag", ["s2", "s1"])
def test_so3_block(golden, tag):
    g = golden(f"module_so3block_{tag}.npz")
    cfg = json.loads(str(g["cfg"]))
    c = golden("constants.npz")
    kp = c["kp24_raw"]
    r = np.sqrt((kp ** 2).sum(1).max())
    seed = int(g["seed"])
    sd = {"inter_conv.conv.anchors": torch.from_numpy(c["anchors"]), "intra_conv.conv.intra_idx": torch.from_numpy(c["intra_idx"]),
          "inter_conv.conv.kernels": torch.from_numpy((kp * (0.7 * cfg["radius"]) / r).astype(np.float32))}
    ci, co = cfg["dim_in"], cfg["dim_out"]
    for name, shape in (("inter_conv.conv.basic_conv.W", (co, ci * 24)), ("inter_conv.conv.basic_conv.bias", (1, co, 1)),
                        ("intra_conv.conv.basic_conv.W", (co, co * 12)), ("intra_conv.conv.basic_conv.bias", (1, co, 1)),
                        ("skip_conv.weight", (co, ci, 1, 1)), ("skip_conv.bias", (co,))):
        sd[name] = seeded_tensor(name, shape, torch.float32, seed)
    xyz, feats, sidx, ball = S.separable_block(sd, "", torch.from_numpy(g["xyz"]), torch.from_numpy(g["feats"]), cfg)
    assert np.array_equal(ball.numpy(), g["ball_idx"])
    if cfg["stride"] > 1:
        assert np.array_equal(sidx.numpy(), g["sample_idx"])
    assert np.array_equal(xyz.numpy(), g["out_xyz"])
    assert np.abs(feats.numpy() - g["out_feats"]).max() < 2e-5


def test_direction_head_and_so3_mean(golden):
    g = golden("module_direction.npz")
    c = golden("constants.npz")
    sd = {k: v for k, v in model_state_dict(int(g["seed"]), c).items() if k.startswith(("direction_", "so3_reg"))}
    w = S.direction_anchor_weights(sd, torch.from_numpy(g["equiv_feat"]))
    assert np.abs(w.numpy() - g["anc_w"]).max() < 1e-5
    R, _, _ = S.so3_mean(torch.from_numpy(c["anchors"]), torch.from_numpy(g["mean_w"]))
    assert np.abs(R.numpy() - g["mean_R"]).max() < 1e-5
    assert np.abs(np.linalg.det(R.numpy()) - 1).max() < 1e-5


def test_propagation(golden):
    g = golden("module_propagation.npz")
    out = S.feat_propagation(torch.from_numpy(g["xyz1"]), torch.from_numpy(g["xyz2"]), torch.from_numpy(g["points2"]))
    assert np.abs(out.numpy() - g["out"]).max() < 1e-6


def test_point_transformer_modules(golden):
    g = golden("module_pt.npz")
    seeds = json.loads(str(g["seeds"]))
    p, x, o = torch.from_numpy(g["p"]), torch.from_numpy(g["x"]), torch.from_numpy(g["o"])
    c = 32

    def sd_for(name, shapes):
        return {k: seeded_tensor(k, s, torch.float32, seeds[name]) for k, s in shapes.items()}

    def bn(pre, n):
        return {pre + "weight": (n,), pre + "bias": (n,), pre + "running_mean": (n,), pre + "running_var": (n,)}

    layer_shapes = {"linear_q.weight": (c, c), "linear_q.bias": (c,), "linear_k.weight": (c, c), "linear_k.bias": (c,),
                    "linear_v.weight": (c, c), "linear_v.bias": (c,), "linear_p.0.weight": (3, 3), "linear_p.0.bias": (3,),
                    **bn("linear_p.1.", 3), "linear_p.3.weight": (c, 3), "linear_p.3.bias": (c,), **bn("linear_w.0.", c),
                    "linear_w.2.weight": (c // 8, c), "linear_w.2.bias": (c // 8,), **bn("linear_w.3.", c // 8),
                    "linear_w.5.weight": (c // 8, c // 8), "linear_w.5.bias": (c // 8,)}
    out = S.pt_layer(sd_for("layer", layer_shapes), "", p, x, o, 8)
    assert np.abs(out.numpy() - g["layer_out"]).max() < 1e-5
    block_shapes = {"linear1.weight": (c, c), **bn("bn1.", c), **{"transformer2." + k: v for k, v in layer_shapes.items()},
                    **bn("bn2.", c), "linear3.weight": (c, c), **bn("bn3.", c)}
    out = S.pt_block(sd_for("block", block_shapes), "", p, x, o, 16)
    assert np.abs(out.numpy() - g["block_out"]).max() < 1e-5
    p2, x2, o2 = S.pt_down(sd_for("down", {"linear.weight": (48, 3 + c), **bn("bn.", 48)}), "", p, x, o, 4, 16)
    assert np.array_equal(p2.numpy(), g["down_p"]) and np.array_equal(o2.numpy(), g["down_o"])
    assert np.abs(x2.numpy() - g["down_x"]).max() < 1e-5
    _, x1, _ = S.pt_down(sd_for("down1", {"linear.weight": (48, c), **bn("bn.", 48)}), "", p, x, o, 1, 8)
    assert np.abs(x1.numpy() - g["down1_x"]).max() < 1e-5
    up_shapes = {"linear1.0.weight": (c, c), "linear1.0.bias": (c,), **bn("linear1.1.", c),
                 "linear2.0.weight": (c, 48), "linear2.0.bias": (c,), **bn("linear2.1.", c)}
    out = S.pt_up(sd_for("up", up_shapes), "", [p, x, o], [torch.from_numpy(g["down_p"]), torch.from_numpy(g["down_x"]), torch.from_numpy(g["down_o"])])
    assert np.abs(out.numpy() - g["up_out"]).max() < 1e-5
    uph_shapes = {"linear1.0.weight": (c, 2 * c), "linear1.0.bias": (c,), **bn("linear1.1.", c), "linear2.0.weight": (c, c), "linear2.0.bias": (c,)}
    out = S.pt_up(sd_for("uph", uph_shapes), "", [p, x, o])
    assert np.abs(out.numpy() - g["uph_out"]).max() < 1e-5


def test_whole_model_n1024(golden):
    g = golden("model_n1024.npz")
    c = golden("constants.npz")
    sd = model_state_dict(int(g["seed"]), c)
    out = S.forward(sd, torch.from_numpy(g["points"]), S.build_layer_table(), return_aux=True)
    assert np.array_equal(out["enc_xyz"].numpy(), g["enc_xyz"])
    assert np.abs(out["enc_feats"][:, :, ::16, :].numpy() - g["enc_feats_sub"]).max() < 1e-5
    for k in ("part_labels", "confidences", "magnitude", "anc_w"):
        scale = np.abs(g[k]).max()
        assert np.abs(out[k].numpy() - g[k]).max() <= 1e-5 * max(scale, 1.0), k
    # direction: compare where the 3x3 polar projection is well conditioned
    ok = (out["sv"][..., 2] / out["sv"][..., 0]).numpy() > 0.1
    assert ok.mean() > 0.3
    assert np.abs(out["direction"].numpy() - g["direction"])[ok].max() < 1e-3


def test_stage2_oracle_reproduces_its_fixture(golden):
    """Regression pin of the stage-2 oracle (parity unpinned upstream, see oracle/stage2.py): a short prefix of the LM
    schedule is re-run and must reproduce the committed per-iteration error trace of oracle/gen_fit_fixture.py."""
    from etch_amd import constants as K
    from etch_amd.utils.body_model import SyntheticSMPL
    from oracle import stage2 as S2
    g = golden("fit_oracle.npz")
    bm = SyntheticSMPL(int(g["body_seed"]))
    mv = np.array(list(K.default_markerset().values()))
    trace = []
    S2.fit_smpl(bm, mv, torch.from_numpy(g["markers"][:1]), torch.from_numpy(g["valid"][:1]), steps_stage0=3, steps_stage1=0, trace=trace)
    got = torch.stack(trace[0], 1).numpy()
    assert got.shape == (1, 4)
    assert np.abs(got - g["err_trace"][:1, :4]).max() <= 1e-4 * g["err_trace"][:1, :4].max()
    # the fixture itself: error decreases monotonically in stage 0 and the fit ends close to the generating body
    tr = g["err_trace"]
    assert (np.diff(tr[:, :31], axis=1) <= 1e-6).all() and (tr[:, -1] < 1e-3).all()
    assert (g["v2v_vs_generating"] < 5e-3).all()            # markers carry 2 mm of noise: the fit lands within a few mm of the generating body


def test_rodrigues_vs_in_tree_batch_rodrigues(golden):
    """oracle.stage2.rodrigues against the golden emitted from the reference's in-tree batch_rodrigues
    (src/data_utils/GT_dataloader_mixed.py:29-64; generator: oracle/ref_harness/gen_golden.py::gen_rodrigues): values in fp32
    and fp64, and the Jacobian dR/dtheta (autograd of the reference function, fp64) at theta = 0, tiny, random and |theta| ~ pi."""
    from oracle import stage2 as S2
    g = golden("rodrigues.npz")
    th = torch.from_numpy(g["theta"])
    assert np.array_equal(S2.rodrigues(th).numpy(), g["R_fp32"])                       # same torch ops in the same order
    assert np.abs(S2.rodrigues(th.double()).numpy() - g["R_fp64"]).max() < 1e-15
    J = torch.stack([torch.autograd.functional.jacobian(lambda v: S2.rodrigues(v[None])[0], t) for t in th.double()])
    assert np.abs(J.numpy() - g["dR_fp64"]).max() < 1e-12
    # sanity of the fixture itself: rotations up to the 1e-8 the formula's `theta + 1e-8` leaves in the axis norm (even in
    # fp64), and the generators at theta = 0
    R = g["R_fp64"]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-6 and np.abs(np.linalg.det(R) - 1).max() < 1e-6
    G0 = g["dR_fp64"][0]                                                              # [i, j, q] at theta = 0
    assert np.abs(G0[2, 1, 0] - 1).max() < 1e-7 and np.abs(G0[1, 2, 0] + 1).max() < 1e-7


def test_whole_model_bundled_4ddress_scan_5k(golden):
    """BASELINE configs[0]'s geometry -- the reference's bundled 4D-Dress scan, 5 000 surface points -- through the oracle's
    stage-1 restatement against the reference's own Python run (tests/golden/scan_4ddress_5k.npz)."""
    g = golden("scan_4ddress_5k.npz")
    c = golden("constants.npz")
    sd = model_state_dict(int(g["seed"]), c)
    out = S.forward(sd, torch.from_numpy(g["points"]), S.build_layer_table(), return_aux=True)
    rows = g["rows"]
    for k, sub in (("part_labels", True), ("confidences", False), ("magnitude", False), ("anc_w", True)):
        got = out[k].numpy()[:, rows] if sub else out[k].numpy()
        scale = np.abs(g[k]).max()
        assert np.abs(got - g[k]).max() <= 1e-5 * max(scale, 1.0), k
    assert (out["part_labels"].argmax(-1).numpy() == g["labels"]).mean() > 0.9995


DEPTH_MLPS = [[32, 32], [64, 64], [128, 128], [256, 256]]


def depth_state_dict(layers, seed, consts):
    """State dict of an EPN_layer_num = `layers` model from ITS manifest (emitted by the reference's own constructor) + seed + constants;
    the kernel-point buffers of the deeper blocks follow functional.py:146-157 on the golden's raw kpsphere24 points."""
    table = S.build_layer_table(mlps=DEPTH_MLPS[:layers], strides=[2] * layers)
    kp = consts["kp24_raw"]
    r = np.sqrt((kp ** 2).sum(1).max())
    sd = {}
    for name, shape, dt in json.load(open(os.path.join(GOLDEN, f"state_dict_manifest_l{layers}.json"))):
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "anchors":
            sd[name] = torch.from_numpy(consts["anchors"])
        elif leaf == "intra_idx":
            sd[name] = torch.from_numpy(consts["intra_idx"])
        elif leaf == "kernels":
            b, c = int(name.split(".")[2]), int(name.split(".")[4])
            sd[name] = torch.from_numpy((kp * (0.7 * table[b][c]["radius"]) / r).astype(np.float32))
        elif leaf == "num_batches_tracked":
            sd[name] = torch.zeros((), dtype=torch.int64)
        else:
            sd[name] = seeded_tensor(name, shape, getattr(torch, dt), seed)
    return sd, table


@pytest.mark.parametrize("layers,n", [(1, 1024), (3, 1024), (4, 512)])
def test_whole_model_other_encoder_depths(golden, layers, n):
    """EPN_layer_num 1 / 3 / 4 (models_pointcloud.py:34-48: feature widths 32 / 128 / 256): the oracle's restatement with the depth's layer
    table against the reference's own Python run of that depth (tests/golden/model_l<d>_n<n>.npz, gen_golden.gen_model_depth)."""
    g, c = golden(f"model_l{layers}_n{n}.npz"), golden("constants.npz")
    assert int(g["layers"]) == layers
    ref_table = json.load(open(os.path.join(GOLDEN, f"epn_model_setting_l{layers}.json")))
    sd, table = depth_state_dict(layers, int(g["seed"]), c)
    for bi, block in enumerate(ref_table["backbone"]):
        for ci, conv in enumerate(block):
            for k, v in table[bi][ci].items():
                assert conv["args"][k] == v, (bi, ci, k)
    assert sd[f"encoder.backbone.{layers - 1}.blocks.1.inter_conv.conv.basic_conv.W"].shape[0] == DEPTH_MLPS[layers - 1][0]
    out = S.forward(sd, torch.from_numpy(g["points"]), table, return_aux=True)
    rows = g["rows"]
    for k, sub in (("part_labels", True), ("confidences", False), ("magnitude", False), ("anc_w", True)):
        got = out[k].numpy()
        got = got[:, rows] if sub else got
        scale = np.abs(g[k]).max()
        assert np.abs(got - g[k]).max() <= 1e-5 * max(scale, 1.0), k
    assert np.array_equal(out["part_labels"].argmax(-1).numpy(), g["labels"])
