"""Round-4 GPU tests: the inter conv with both contractions on the bf16 matrix cores (csrc/so3conv_x.hip), the bf16-plane producers."""
import numpy as np
import pytest
import torch

from etch_amd.utils.weights import load_seeded
from tests.test_gpu_encoder import _inter_conv_fp64, rel_err

pytestmark = pytest.mark.gpu


def test_split3_planes_are_exact_and_match_the_host_split():
    """etch_split3_planes / instnorm_act_add(want_planes=True): the three bf16 planes sum to the fp32 value exactly and equal ops.split3_bf16."""
    from etch_amd import ops
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(2, 37, 60, 32, generator=g) * torch.logspace(-6, 6, 32)).cuda()
    planes = ops.split3_planes(x)
    assert planes.shape == (2, 37, 60, 3, 32) and planes.dtype == torch.int16
    host = ops.split3_bf16(x)                                   # [3][...]
    assert torch.equal(planes, host.permute(1, 2, 3, 0, 4))
    back = sum((planes[..., k, :].to(torch.int32) << 16).view(torch.float32).double() for k in range(3))
    assert torch.equal(back.float(), x) and torch.equal(back, x.double())
    m, r = ops.instnorm_stats(x)
    x2 = torch.randn(2, 37, 60, 32, generator=g).cuda()
    m2, r2 = ops.instnorm_stats(x2)
    out = ops.instnorm_act_add(x, m, r, x2, m2, r2)
    out_p, pl = ops.instnorm_act_add(x, m, r, x2, m2, r2, want_planes=True)
    assert torch.equal(out, out_p) and torch.equal(pl, ops.split3_planes(out))


@pytest.mark.parametrize("form", [32, 16])
@pytest.mark.parametrize("cin,cout,nn,p1,p2", [(32, 32, 32, 301, 149), (32, 64, 64, 211, 60), (64, 64, 32, 211, 101), (32, 32, 64, 130, 1),
                                                (32, 64, 32, 97, 97), (64, 64, 64, 150, 33)])
def test_inter_conv_planes_matches_the_fp32_kernel_and_fp64(cin, cout, nn, p1, p2, form):
    """etch_inter_so3conv_planes32 / etch_inter_so3conv_planes (32x32x16 / 16x16x32 MFMA form; BOTH contractions on the bf16 matrix cores: fp32 operands split exactly into three bf16 values, six cross
    products each; gathered rows as producer-written planes through LDS-direct loads + transposing LDS reads) against the fp32-MFMA kernel
    (the same sums in another order) and the fp64 formula under the entitled-error rule; bitwise reproducible, schedule-independent;
    padded neighbourhoods (radius smaller than the cloud: cyclic padding) included."""
    from etch_amd import _lib, ops
    from etch_amd import vgtk_so3conv as V
    if not _lib.has_experiments():
        pytest.skip("the round-4 planes kernels are built only with ETCH_BUILD_EXPERIMENTS=1 since round 6 (VERDICT r05 item 9)")
    assert ops.inter_planes_supported(cin, cout, nn)
    g = torch.Generator().manual_seed(cin + nn + p2)
    b = 2
    xyz = (torch.randn(b, 3, p1, generator=g) * 0.2).cuda()
    new_xyz = xyz[:, :, :p2].contiguous()
    ball = ops.ball_query(new_xyz, xyz, 0.25, nn)
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, 2, 0.25, 0.03, nn), 3).cuda()
    rk, W, Wp, bias = conv._derived()
    if form == 32 and cin != 64:
        pytest.skip("the 32x32x16 form is instantiated for 64 input channels only")
    wkw = dict(Wq32=conv._wq32()) if form == 32 else dict(Wqn=conv._wqn())
    assert all(v is not None for v in wkw.values())
    feats = torch.randn(b, p1, 60, cin, generator=g).cuda()
    planes = ops.split3_planes(feats)
    f32, (m0, r0) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, want_stats=True)
    new, (m1, r1) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, want_stats=True, feats_planes=planes,
                                      order=ops.spatial_order(new_xyz), **wkw)
    again = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, **wkw)        # planes made on the fly, plain order
    assert torch.equal(new, again)
    scale = float(f32.abs().max())
    assert float((new - f32).abs().max()) < 2e-6 * scale, float((new - f32).abs().max()) / scale
    assert rel_err(m1.cpu().numpy(), m0.cpu().numpy()) < 2e-6 and rel_err(r1.cpu().numpy(), r0.cpu().numpy()) < 2e-6
    ref = _inter_conv_fp64(xyz, new_xyz, ball, feats, rk, W, bias, conv.sigma)
    e_new, e_f32 = float((new.double() - ref).abs().max()), float((f32.double() - ref).abs().max())
    assert e_f32 < 3e-6 * scale and e_new <= 2.0 * e_f32 + 1e-7 * scale, (e_new, e_f32)


def test_encoder_with_plane_producers_equals_encoder_without(monkeypatch):
    """The encoder with its blocks emitting planes for the next conv == the encoder that splits inside ops.inter_so3conv (to the planes' last bit), and both
    within the split kernels' bound of the round-3 path (ETCH_INTER_X=0: step 1 on the fp32 MFMA)."""
    import types

    from etch_amd import constants as K
    from etch_amd import ops
    from etch_amd.models.models_pointcloud import GT_network_equiv
    args = types.SimpleNamespace(output_folder="/tmp/etch_r04", EPN_input_radius=0.4, EPN_layer_num=2, device=torch.device("cuda"),
                                 markerset=K.default_markerset())
    model = load_seeded(GT_network_equiv(option=args), 1).cuda().eval()
    emit = [c.emit_planes for blk in model.encoder.backbone for c in blk.blocks]
    assert [bool(e) for e in emit] == [True, True, True, False], emit
    pts = torch.from_numpy((np.random.default_rng(5).standard_normal((2, 1500, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)).cuda()
    with torch.no_grad():
        a = model.encoder(pts)[0].feats_cl
        for blk in model.encoder.backbone:
            for c in blk.blocks:
                c.emit_planes = False
        bb = model.encoder(pts)[0].feats_cl
        # (bitwise until round 5; since round 6 planes made inside ops.inter_so3conv carry every scan's own power of two -- the same sums from operands
        # shifted by a power of two: equal up to the planes' last bit, through four convs)
        assert float((a - bb).abs().max()) < 2e-6 * float(a.abs().max())
        monkeypatch.setattr(ops, "INTER_X", False)
        c3 = model.encoder(pts)[0].feats_cl
    assert float((a - c3).abs().max()) < 2e-5 * float(c3.abs().max())


def test_split_lm_fit_gives_up_loudly_when_a_partner_never_arrives():
    """the split LM fit must not go on with partial tiles.  With one workgroup of every scan dropped at launch (test hook) the others time
    out, flag the scan, and the scan's parameters / trace come back NaN with the flag counted; the next, normal call is unaffected; and a batch whose
    B * G exceeds the CU count falls back to one workgroup per scan instead of raising."""
    from etch_amd import _lib, ops
    from etch_amd.models.fit_SMPL import _device_body
    from etch_amd import constants as K
    from etch_amd.utils.body_model import SyntheticSMPL
    import bench as Bn
    import types
    dev = torch.device("cuda")
    args = types.SimpleNamespace(body_model=SyntheticSMPL(7), markerset=K.default_markerset())
    mk, vf, vb, _ = Bn.well_posed_markers(args, dev, 4)
    db = _device_body(args.body_model, np.array(list(args.markerset.values())), dev)
    good, _, tr_good = ops.smpl_lm_fit(db.lm_consts, mk, vf, 6, 0.5, 0.01, 6, 0.2, 1e-3, want_trace=True, split=3)
    assert ops.smpl_lm_split_failed(good) == 0 and bool(torch.isfinite(good).all())
    lib = _lib.lib()
    try:
        lib.etch_smpl_lm_debug(2000, 1)                                  # ~ a millisecond of polling; group 1 of every scan never runs
        bad, bad0, tr_bad = ops.smpl_lm_fit(db.lm_consts, mk, vf, 6, 0.5, 0.01, 6, 0.2, 1e-3, want_trace=True, split=3)
        assert ops.smpl_lm_split_failed(bad) == 4
        assert bool(torch.isnan(bad).all()) and bool(torch.isnan(bad0).all()) and bool(torch.isnan(tr_bad).all())
    finally:
        lib.etch_smpl_lm_debug(0, -1)
    again, _, tr_again = ops.smpl_lm_fit(db.lm_consts, mk, vf, 6, 0.5, 0.01, 6, 0.2, 1e-3, want_trace=True, split=3)
    assert torch.equal(again, good) and torch.equal(tr_again, tr_good)
    # more workgroups than CUs: falls back to G = 1 (no exception), same fit up to the summation order of the normal matrix
    big = 100
    mkb, vfb = mk.repeat(big // 4, 1, 1), vf.repeat(big // 4, 1)
    xb, _, _ = ops.smpl_lm_fit(db.lm_consts, mkb, vfb, 6, 0.5, 0.01, 6, 0.2, 1e-3, split=3)
    assert getattr(xb, "_etch_split_ws", None) is None
    assert float((xb[:4] - good).abs().max()) < 1e-4


def _full_vs_shards(tmp_path, total, shard, n, body, iters):
    """The 8-GPU job's WHOLE batch on one GPU against two of its per-GPU shards (first and last): a scan's result must not depend on its batch."""
    import types

    from etch_amd import constants as K
    from etch_amd.inference_demo import predict_smpl_batch
    from etch_amd.models.models_pointcloud import GT_network_equiv
    items = ["confidence", "direction", "magnitude"]
    args = types.SimpleNamespace(output_folder=str(tmp_path), EPN_input_radius=0.4, EPN_layer_num=2, device=torch.device("cuda"),
                                 markerset=K.default_markerset(), scale_magnitude=10, body_model=body)
    model = load_seeded(GT_network_equiv(option=args), 1).cuda().eval()
    scan = lambda s: (np.random.default_rng(s).standard_normal((n, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)
    pts = torch.from_numpy(np.stack([scan(1000 + b) for b in range(total)])).cuda()
    kw = dict(steps_stage0=iters[0], steps_stage1=iters[1])
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        res, _ = model(pts, items, "standard_vector")
    meshes, markers, valid, info, aux = predict_smpl_batch(args, model, pts, "neutral", return_trace=True, **kw)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    for k, v in res.items():
        assert bool(torch.isfinite(v.float()).all()), k
    # (a scan whose marker weights conf**20 underflow carries a NaN marker, exactly as fit_SMPL.py:52-57 -- status bit 0; seeded random weights produce a few)
    ok = (aux["status"] == 0).cpu().numpy()
    assert ok.sum() >= total // 2 and len(meshes) == total
    assert bool(torch.isfinite(aux["err_trace"][torch.from_numpy(ok).cuda()]).all()) and all(np.isfinite(a[ok]).all() for a in info)
    for first in (0, total - shard):
        sub = pts[first:first + shard].contiguous()
        with torch.no_grad():
            r2, _ = model(sub, items, "standard_vector")
        _, mk2, va2, info2, aux2 = predict_smpl_batch(args, model, sub, "neutral", return_trace=True, **kw)
        for s_full, s_sub in ((first, 0), (first + shard - 1, shard - 1)):
            for k in ("part_labels", "direction", "magnitude"):
                assert torch.equal(res[k][s_full], r2[k][s_sub]), (k, s_full)
            # confidences: the fused head's 128-row tiles start at different rows -> the last bit may differ with the scan's position in the batch
            assert float((res["confidences"][s_full] - r2["confidences"][s_sub]).abs().max()) <= 1e-7
            assert torch.equal(valid[s_full], va2[s_sub])
            m_ok = torch.isfinite(markers[s_full]).all(-1)
            assert float((markers[s_full][m_ok] - mk2[s_sub][m_ok]).abs().max()) <= 1e-6
            if ok[s_full]:
                assert np.abs(info[0][s_full] - info2[0][s_sub]).max() < 1e-4 and np.abs(info[4][s_full] - info2[4][s_sub]).max() < 1e-4
    return peak


def test_config3_whole_8gpu_batch_256x5000_on_one_gpu(tmp_path):
    """BASELINE configs[3] (256 scans x 5 000 points, full pipeline) as ONE batch on one MI355X: finite everywhere, scans 0 / 31 / 224 / 255 equal to
    their 32-scan per-GPU shards (bitwise but for the confidence head's last bit), peak HBM reported."""
    from etch_amd.utils.body_model import SyntheticSMPL
    peak = _full_vs_shards(tmp_path, 256, 32, 5000, SyntheticSMPL(7), (30, 50))
    print(f"configs[3] on one GPU: peak HBM {peak:.1f} GiB")
    assert peak < 200


def test_config4_whole_8gpu_batch_64x20000_smplx_on_one_gpu(tmp_path):
    """BASELINE configs[4] (64 dense 20 000-point scans + 75+125-iteration fit of the 188-DoF body) as ONE batch on one MI355X against its 8-scan shards."""
    from etch_amd.utils.body_model import SyntheticSMPLX
    peak = _full_vs_shards(tmp_path, 64, 8, 20000, SyntheticSMPLX(7), (75, 125))
    print(f"configs[4] on one GPU: peak HBM {peak:.1f} GiB")
    assert peak < 200


def test_selfcheck_compares_a_dump_section_by_section(tmp_path):
    """etch_amd.selfcheck (pinning the CUDA index kernels from a dump made on a CUDA box): here the dump is written by the CPU
    oracle, so every section must match; a corrupted entry must be reported as a mismatch."""
    from etch_amd import selfcheck
    from oracle import ops as O
    rng = np.random.default_rng(3)
    xyz = (rng.standard_normal((2, 3, 700)) * 0.2).astype(np.float32)
    new_xyz = xyz[:, :, :200].copy()
    pk = (rng.standard_normal((900, 3)) * 0.3).astype(np.float32)
    off, noff = np.array([400, 900], np.int32), np.array([100, 225], np.int32)
    fidx = O.furthestsampling(pk, off, noff)
    kidx, kd2 = O.knnquery(16, pk, pk[fidx], off, noff)
    dump = dict(fps_xyz=xyz, fps_m=350, fps_idx=O.furthest_point_sampling(xyz, 350), bq_new_xyz=new_xyz, bq_xyz=xyz, bq_radius=0.15, bq_nsample=32,
                bq_idx=O.ball_query(new_xyz, xyz, 0.15, 32), pfps_xyz=pk, pfps_offset=off, pfps_new_offset=noff, pfps_idx=fidx,
                knn_xyz=pk, knn_new_xyz=pk[fidx], knn_offset=off, knn_new_offset=noff, knn_nsample=16, knn_idx=kidx, knn_dist=np.sqrt(kd2))
    path = tmp_path / "dump.npz"
    np.savez(path, **dump)
    rep = selfcheck.check(dict(np.load(path)))
    assert set(rep) == {"vgtk_fps", "ball_query", "pointops_fps", "pointops_knn_idx", "pointops_knn_dist"} and all(v["match"] for v in rep.values()), rep
    dump["bq_idx"] = dump["bq_idx"].copy()
    dump["bq_idx"][1, 7, 3] += 1
    rep = selfcheck.check(dump)
    assert rep["ball_query"]["match"] is False and rep["ball_query"]["differing"] == 1 and rep["vgtk_fps"]["match"]
