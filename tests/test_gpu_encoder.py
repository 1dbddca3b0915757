"""GPU parity of the dense building blocks and the EPN encoder vs golden vectors
from the reference and vs the oracle.  Tolerance: 1e-4 relative to the tensor's max (fp32, north_star)."""
import json

import numpy as np
import pytest
import torch

from etch_amd.utils.weights import load_seeded

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize("R,K,O", [(1000, 64, 64), (777, 128, 86), (64, 32, 16), (5000, 131, 128), (300, 3, 35), (129, 67, 128),
                                   (2048, 512, 512), (100, 1, 32),
                                   # the weight-stationary split kernel's shapes (gemm_ws_split_kernel): ragged row blocks, one row, every (SPW, RP)
                                   (3000, 128, 128), (500, 128, 384), (777, 32, 32), (1000, 64, 192), (130, 32, 64), (4097, 64, 256), (900, 128, 256),
                                   (1, 64, 128), (20000, 32, 128)])
def test_linear_epilogues(R, K, O):
    from etch_amd import ops
    rng = np.random.default_rng(R + K)
    x = rng.standard_normal((R, K)).astype(np.float32)
    w = (rng.standard_normal((O, K)) / np.sqrt(K)).astype(np.float32)
    bias, sc, sh = (rng.standard_normal(O).astype(np.float32) for _ in range(3))
    res = rng.standard_normal((R, O)).astype(np.float32)
    d = lambda a: torch.from_numpy(a).cuda()
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    y = ops.linear(d(x), d(w)).cpu().numpy()
    assert rel_err(y, ref) < 1e-5
    y = ops.linear(d(x), d(w), bias=d(bias), scale=d(sc), shift=d(sh), act="relu").cpu().numpy()
    assert rel_err(y, np.maximum((ref + bias) * sc + sh, 0)) < 1e-5
    y = ops.linear(d(x), d(w), bias=d(bias), act="relu", res=d(res), res_mode=1).cpu().numpy()
    assert rel_err(y, np.maximum(ref + bias + res, 0)) < 1e-5
    y = ops.linear(d(x), d(w), act="leaky_relu", res=d(res), res_mode=2).cpu().numpy()
    assert rel_err(y, np.where(ref > 0, ref, 0.01 * ref) + res) < 1e-5


def test_linear_grouped_row_gather():
    from etch_amd import ops
    rng = np.random.default_rng(0)
    b, p_in, p_out, grp, K, O = 3, 50, 20, 60, 32, 48
    x = rng.standard_normal((b, p_in, grp, K)).astype(np.float32)
    w = rng.standard_normal((O, K)).astype(np.float32)
    sidx = np.stack([rng.permutation(p_in)[:p_out] for _ in range(b)]).astype(np.int32)
    y = ops.linear(torch.from_numpy(x).cuda().view(-1, K), torch.from_numpy(w).cuda(), row_idx=torch.from_numpy(sidx).cuda(),
                   grp=grp, p_in=p_in, p_out=p_out, rows=b * p_out * grp).cpu().numpy().reshape(b, p_out, grp, O)
    ref = np.stack([x[i, sidx[i]] for i in range(b)]).astype(np.float64) @ w.astype(np.float64).T
    assert rel_err(y, ref) < 1e-5


def test_instnorm(golden):
    from etch_amd import ops
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((3, 700, 60, 32)) * 3 + 1.5).astype(np.float32)
    x2 = rng.standard_normal((3, 700, 60, 32)).astype(np.float32)
    xt, x2t = torch.from_numpy(x).cuda(), torch.from_numpy(x2).cuda()
    m, r = ops.instnorm_stats(xt)
    xd = x.astype(np.float64).reshape(3, -1, 32)
    assert np.abs(m.cpu().numpy() - xd.mean(1)).max() < 1e-6
    assert rel_err(r.cpu().numpy(), 1 / np.sqrt(xd.var(1) + 1e-5)) < 1e-6
    lre = lambda v: np.where(v > 0, v, 0.01 * v)
    n1 = lre((xd - xd.mean(1, keepdims=True)) / np.sqrt(xd.var(1, keepdims=True) + 1e-5))
    y = ops.instnorm_act_add(xt, m, r).cpu().numpy().reshape(3, -1, 32)
    assert rel_err(y, n1) < 1e-5
    m2, r2 = ops.instnorm_stats(x2t)
    x2d = x2.astype(np.float64).reshape(3, -1, 32)
    n2 = lre((x2d - x2d.mean(1, keepdims=True)) / np.sqrt(x2d.var(1, keepdims=True) + 1e-5))
    y = ops.instnorm_act_add(xt, m, r, x2t, m2, r2).cpu().numpy().reshape(3, -1, 32)
    assert rel_err(y, n1 + n2) < 1e-5


@pytest.mark.parametrize("tag", ["s2", "s1"])
def test_separable_block_vs_reference_golden(golden, tag):
    from etch_amd import vgtk_so3conv as sptk
    from etch_amd.models.so3conv import SeparableSO3ConvBlock
    g = golden(f"module_so3block_{tag}.npz")
    cfg = json.loads(str(g["cfg"]))
    params = dict(cfg, kernel_size=1, dropout_rate=0, multiplier=2, activation="leaky_relu", pooling=None, kanchor=60)
    blk = load_seeded(SeparableSO3ConvBlock(params), int(g["seed"])).cuda().eval()
    x = sptk.SphericalPointCloud(torch.from_numpy(g["xyz"]).cuda(), torch.from_numpy(g["feats"]).cuda(), None)
    idx, _, sidx, out = blk(x, None, None)
    assert np.array_equal(idx.cpu().numpy(), g["ball_idx"])
    if cfg["stride"] > 1:
        assert np.array_equal(sidx.cpu().numpy(), g["sample_idx"])
    assert np.array_equal(out.xyz.cpu().numpy(), g["out_xyz"])
    assert out.feats.shape == g["out_feats"].shape
    assert rel_err(out.feats.cpu().numpy(), g["out_feats"]) < RTOL


def test_encoder_vs_reference_golden(golden):
    from etch_amd.config.EPN_options import get_default_cfg
    from etch_amd.models.so3net import build_model
    g = golden("model_n1024.npz")
    cfg = get_default_cfg()
    enc = build_model(cfg, mlps=[[32, 32], [64, 64]], strides=[2, 2])
    # seeded weights are keyed by the full model's names: prefix "encoder."
    from etch_amd.utils.weights import seeded_tensor
    sd = enc.state_dict()
    for k, v in sd.items():
        if not k.endswith(("anchors", "kernels", "intra_idx")):
            sd[k] = seeded_tensor("encoder." + k, v.shape, v.dtype, int(g["seed"]))
    enc.load_state_dict(sd)
    enc = enc.cuda().eval()
    out, sample_lists = enc(torch.from_numpy(g["points"]).cuda())
    assert np.array_equal(out.xyz.cpu().numpy(), g["enc_xyz"])
    f = out.feats.cpu().numpy()
    assert f.shape == (2, 64, 256, 60)
    assert rel_err(f[:, :, ::16, :], g["enc_feats_sub"]) < RTOL


def _morton_reference(x, bits=10, base=0, count=None):
    """numpy restatement of etch_spatial_order: `bits` per axis on the scan's bounding box, ties by index; for scans of more than
    16 384 points: 5 bits per axis and independent 32 768-point slices [base, base + count)."""
    x = x.astype(np.float32)
    cells = np.float32(2 ** bits - 1)
    lo, hi = x.min(1, keepdims=True), x.max(1, keepdims=True)
    sc = (cells / np.where(hi > lo, hi - lo, np.float32(1.0)) * (hi > lo)).astype(np.float32)
    q = np.clip(((x - lo) * sc).astype(np.float32), 0, cells).astype(np.uint64)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    key = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2)
    count = x.shape[1] - base if count is None else count
    idx = np.arange(base, base + count)
    return idx[np.lexsort((idx, key[idx]))]


@pytest.mark.parametrize("b,n", [(3, 2500), (2, 1250), (1, 1), (2, 4097), (1, 16384)])
def test_spatial_order_is_the_morton_permutation(b, n):
    from etch_amd import ops
    rng = np.random.default_rng(n)
    xyz = (rng.standard_normal((b, 3, n)) * np.array([0.14, 0.31, 0.085])[None, :, None]).astype(np.float32)
    if n > 10:
        xyz[0, :, 5] = xyz[0, :, 3]                      # a duplicated point: tie broken by index
    order = ops.spatial_order(torch.from_numpy(xyz).cuda()).cpu().numpy()
    assert order.shape == (b, n) and order.dtype == np.int32
    for i in range(b):
        assert np.array_equal(np.sort(order[i]), np.arange(n))
        assert np.array_equal(order[i], _morton_reference(xyz[i]))


@pytest.mark.parametrize("b,n", [(2, 20000), (1, 16385), (1, 32768), (1, 40001)])
def test_spatial_order_beyond_one_lds_sort(b, n):
    """BASELINE configs[4] sizes (20 000 points per scan): 15-bit codes, 32 768-point slices -- still a permutation of every scan, and
    exactly the coarse Morton walk of each slice."""
    from etch_amd import ops
    rng = np.random.default_rng(n)
    xyz = (rng.standard_normal((b, 3, n)) * np.array([0.14, 0.31, 0.085])[None, :, None]).astype(np.float32)
    order = ops.spatial_order(torch.from_numpy(xyz).cuda()).cpu().numpy()
    assert order.shape == (b, n) and order.dtype == np.int32
    for i in range(b):
        assert np.array_equal(np.sort(order[i]), np.arange(n))
        want = np.concatenate([_morton_reference(xyz[i], 5, s, min(32768, n - s)) for s in range(0, n, 32768)])
        assert np.array_equal(order[i], want)


def test_inter_conv_result_does_not_depend_on_the_schedule(golden):
    """The spatial schedule only permutes which workgroup computes which output point: bitwise identical outputs, also with an
    arbitrary permutation and with a point count that is not a multiple of 8 (empty slots of the XCD-striped grid)."""
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(5)
    b, p1, p2, nn, cin, cout = 2, 301, 149, 32, 32, 64
    xyz = (torch.randn(b, 3, p1, generator=g) * 0.2).cuda()
    new_xyz = xyz[:, :, :p2].contiguous()
    ball = ops.ball_query(new_xyz, xyz, 0.25, nn)
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, 2, 0.25, 0.03, nn), 3).cuda()
    rk, W, Wp, bias = conv._derived()
    feats = torch.randn(b, p1, 60, cin, generator=g).cuda()
    ref = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma)
    for order in (ops.spatial_order(new_xyz), torch.stack([torch.randperm(p2, generator=g) for _ in range(b)]).int().cuda()):
        out = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, order=order)
        assert torch.equal(out, ref)


@pytest.mark.parametrize("cin,cout,nn", [(32, 64, 32), (64, 64, 20), (1, 32, 64)])
def test_inter_conv_fused_instancenorm_statistics(cin, cout, nn):
    """mean / rstd accumulated in the conv's epilogue (etch_inter_so3conv_ordered stat_part + etch_instnorm_from_partials) equal the
    separate statistics pass over the written output (so3conv.py:96-99 InstanceNorm2d) to fp32 rounding; the output itself is untouched."""
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(cin + nn)
    b, p1, p2 = 3, 211, 101
    xyz = (torch.randn(b, 3, p1, generator=g) * 0.2).cuda()
    new_xyz = xyz[:, :, :p2].contiguous()
    ball = ops.ball_query(new_xyz, xyz, 0.25, nn)
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, 2, 0.25, 0.03, nn), 3).cuda()
    rk, W, Wp, bias = conv._derived()
    feats = torch.randn(b, p1, 60, cin, generator=g).cuda()
    ref = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma)
    out, (m, r) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, order=conv.order(new_xyz), want_stats=True)
    assert torch.equal(out, ref)
    m0, r0 = ops.instnorm_stats(ref)
    assert rel_err(m.cpu().numpy(), m0.cpu().numpy()) < 2e-6 and rel_err(r.cpu().numpy(), r0.cpu().numpy()) < 2e-6
    x64 = ref.double().reshape(b, -1, cout)
    assert rel_err(m.cpu().numpy(), x64.mean(1).cpu().numpy()) < 2e-6
    assert rel_err(r.cpu().numpy(), (1.0 / torch.sqrt(x64.var(1, unbiased=False) + 1e-5)).cpu().numpy()) < 2e-6


@pytest.mark.parametrize("c,p,normed", [(32, 100, True), (64, 58, False), (16, 6, True), (32, 51, True)])
def test_intra_conv_fused_instancenorm_statistics(c, p, normed):
    """mean / rstd from the intra conv's epilogue (etch_intra_so3conv_stats + etch_instnorm_from_partials) equal the separate statistics
    pass over the written output; the output itself is untouched; odd point counts take the separate pass."""
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(c + p)
    b = 3
    conv = load_seeded(V.IntraSO3Conv(c, c), 5).cuda()
    Wp, bias, idx32, Wp32 = conv._derived()
    x = (torch.randn(b, p, 60, c, generator=g) * 2 + 0.5).cuda()
    m1, r1 = ops.instnorm_stats(x) if normed else (None, None)
    ref = ops.intra_so3conv(x, idx32, Wp, bias, c, m1, r1, Wp32=Wp32)
    out, (m, r) = ops.intra_so3conv(x, idx32, Wp, bias, c, m1, r1, want_stats=True, Wp32=Wp32)
    assert torch.equal(out, ref)
    if Wp32 is not None:
        # the 32x32x2 MFMA kernel against the 16x16x4 one: same function, a different split of the fp32 sums over the lane groups
        old = ops.intra_so3conv(x, idx32, Wp, bias, c, m1, r1)
        assert rel_err(out.cpu().numpy(), old.cpu().numpy()) < 2e-6
    x64 = ref.double().reshape(b, -1, c)
    assert rel_err(m.cpu().numpy(), x64.mean(1).cpu().numpy()) < 2e-6
    assert rel_err(r.cpu().numpy(), (1.0 / torch.sqrt(x64.var(1, unbiased=False) + 1e-5)).cpu().numpy()) < 2e-6


@pytest.mark.parametrize("kind", ["inter", "intra"])
def test_fused_instancenorm_statistics_with_a_dominant_channel_mean(kind):
    """A channel whose mean dominates its spread (bias of 300 on outputs of spread ~1: mean/std ~ 300): squares formed in fp32 would lose
    ~ (mean/std)^2 * 6e-8 = 5e-3 of the variance; the epilogue accumulates them in fp64 and must match the fp64 statistics of the written
    output to 1e-5 in rstd (what remains is the fp32 rounding of the outputs themselves)."""
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(17)
    b = 2
    if kind == "inter":
        cin, cout, nn, p1, p2 = 32, 32, 24, 180, 90
        xyz = (torch.randn(b, 3, p1, generator=g) * 0.2).cuda()
        new_xyz = xyz[:, :, :p2].contiguous()
        ball = ops.ball_query(new_xyz, xyz, 0.25, nn)
        conv = load_seeded(V.InterSO3Conv(cin, cout, 1, 2, 0.25, 0.03, nn), 3).cuda()
        rk, W, Wp, bias = conv._derived()
        bias = bias.clone()
        bias[::2] += 300.0
        feats = torch.randn(b, p1, 60, cin, generator=g).cuda()
        out, (m, r) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, order=conv.order(new_xyz), want_stats=True)
    else:
        c, p = 32, 64
        cout = c
        conv = load_seeded(V.IntraSO3Conv(c, c), 5).cuda()
        Wp, bias, idx32, Wp32 = conv._derived()
        bias = bias.clone()
        bias[::2] += 300.0
        x = torch.randn(b, p, 60, c, generator=g).cuda()
        out, (m, r) = ops.intra_so3conv(x, idx32, Wp, bias, c, None, None, want_stats=True, Wp32=Wp32)
    x64 = out.double().reshape(b, -1, cout)
    std = x64.std(1, unbiased=False)
    assert float((x64.mean(1).abs() / std).max()) > 100
    assert rel_err(m.cpu().numpy(), x64.mean(1).cpu().numpy()) < 1e-6
    want = (1.0 / torch.sqrt(x64.var(1, unbiased=False) + 1e-5)).cpu().numpy()
    assert np.abs(r.cpu().numpy() / want - 1).max() < 1e-5


def _inter_conv_fp64(xyz, new_xyz, ball, feats, rk, W, bias, sigma):
    """The dense formula of functional.py:286-324 + modules.py:33-39 in fp64 on the device: w = relu(1 - |g - R kappa|^2 / sigma),
    X1[a,k,c] = sum_n w[a,k,n] F[idx_n, a, c], Y = W X1 + bias."""
    b, _, p2 = new_xyz.shape
    nn_, cin = ball.shape[2], feats.shape[3]
    idx = ball.long()
    X = xyz.double().permute(0, 2, 1)                                                    # b, p1, 3
    g = torch.gather(X[:, None].expand(-1, p2, -1, -1), 2, idx[..., None].expand(-1, -1, -1, 3)) - new_xyz.double().permute(0, 2, 1)[:, :, None]
    d2 = ((g[:, :, None, None] - rk.double()[None, None, :, :, None]) ** 2).sum(-1)      # b, p2, 60, 24, nn
    w = torch.clamp(1.0 - d2 / sigma, min=0.0)
    F = torch.gather(feats.double()[:, None].expand(-1, p2, -1, -1, -1), 2, idx[..., None, None].expand(-1, -1, -1, 60, cin))   # b, p2, nn, 60, c
    X1 = torch.einsum("bpakn,bpnac->bpack", w, F).reshape(b, p2, 60, cin * 24)
    return X1 @ W.double().t() + bias.double()


@pytest.mark.parametrize("cin,cout,nn,p1,p2", [(32, 32, 32, 301, 149), (32, 64, 64, 211, 60), (64, 64, 20, 211, 101), (16, 32, 9, 100, 1),
                                                (128, 128, 32, 150, 33)])
def test_inter_conv_split_operands_match_the_fp32_kernel_and_fp64(cin, cout, nn, p1, p2):
    """etch_inter_so3conv_split (step 2 on the bf16 matrix cores, fp32 operands split exactly into three bf16 values, six cross products) against
    the fp32-MFMA kernel (same sums in a different order: <= 2e-6 of the output's scale) and against the fp64 formula: it may not be further
    from fp64 than twice the fp32 kernel is (the entitled-error rule of the parity tests); bitwise reproducible, schedule-independent."""
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(cin + nn)
    b = 2
    xyz = (torch.randn(b, 3, p1, generator=g) * 0.2).cuda()
    new_xyz = xyz[:, :, :p2].contiguous()
    ball = ops.ball_query(new_xyz, xyz, 0.25, nn)
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, 2, 0.25, 0.03, nn), 3).cuda()
    rk, W, Wp, bias = conv._derived()
    Wq = conv._wq()
    feats = torch.randn(b, p1, 60, cin, generator=g).cuda()
    f32, (m0, r0) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, want_stats=True)
    new, (m1, r1) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, want_stats=True, Wq=Wq, order=ops.spatial_order(new_xyz))
    again = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, Wq=Wq)
    assert torch.equal(new, again)
    scale = float(f32.abs().max())
    assert float((new - f32).abs().max()) < 2e-6 * scale
    assert rel_err(m1.cpu().numpy(), m0.cpu().numpy()) < 2e-6 and rel_err(r1.cpu().numpy(), r0.cpu().numpy()) < 2e-6
    ref = _inter_conv_fp64(xyz, new_xyz, ball, feats, rk, W, bias, conv.sigma)
    e_new, e_f32 = float((new.double() - ref).abs().max()), float((f32.double() - ref).abs().max())
    assert e_f32 < 3e-6 * scale and e_new <= 2.0 * e_f32 + 1e-7 * scale, (e_new, e_f32)


@pytest.mark.parametrize("c,p,b,normed", [(64, 256, 2, True), (32, 512, 2, False), (64, 57, 3, True), (32, 1, 1, True), (64, 2, 1, False)])
def test_intra_conv_weight_stationary_split_matches_the_fp32_kernel_and_fp64(c, p, b, normed):
    """etch_intra_so3conv_split (weight-stationary, bf16 matrix cores, exactly split fp32 operands) against the fp32 kernels and the fp64
    formula (functional.py:331-378 + modules.py:150-153, InstanceNorm + LeakyReLU on load as so3conv.py:96-99); odd point counts, one point."""
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(c + p)
    conv = load_seeded(V.IntraSO3Conv(c, c), 5).cuda()
    Wp, bias, idx32, Wp32 = conv._derived()
    x = (torch.randn(b, p, 60, c, generator=g) * 2 + 0.5).cuda()
    mm, rr = ops.instnorm_stats(x) if normed else (None, None)
    ws = p % 2 == 0
    new = ops.intra_so3conv(x, idx32, Wp, bias, c, mm, rr, Wp32=Wp32, Wq=conv._wq, want_stats=ws)
    f32 = ops.intra_so3conv(x, idx32, Wp, bias, c, mm, rr, want_stats=ws)
    if ws:
        (new, (m1, r1)), (f32, (m0, r0)) = new, f32
        assert rel_err(m1.cpu().numpy(), m0.cpu().numpy()) < 2e-6 and rel_err(r1.cpu().numpy(), r0.cpu().numpy()) < 2e-6
    assert torch.equal(new, ops.intra_so3conv(x, idx32, Wp, bias, c, mm, rr, Wq=conv._wq))
    scale = float(f32.abs().max())
    assert float((new - f32).abs().max()) < 3e-6 * scale
    xd = x.double()
    if normed:
        xd = (xd - mm.double()[:, None, None]) * rr.double()[:, None, None]
        xd = torch.where(xd > 0, xd, 0.01 * xd)
    gathered = xd[:, :, conv.intra_idx.cuda()]                                           # b, p, 60, 12, c
    W3 = conv.basic_conv.W.detach().double().view(c, c, 12)                              # [o][ch][tap]
    ref = torch.einsum("bpatc,oct->bpao", gathered, W3) + bias.double()
    e_new, e_f32 = float((new.double() - ref).abs().max()), float((f32.double() - ref).abs().max())
    assert e_f32 < 3e-6 * scale and e_new <= 2.0 * e_f32 + 1e-7 * scale, (e_new, e_f32)


@pytest.mark.parametrize("cin,cout,nn", [(32, 32, 32), (32, 64, 64), (64, 64, 20)])
def test_inter_conv_32x32x2_variant_matches(cin, cout, nn):
    """The opt-in two-points-per-workgroup kernel on v_mfma_f32_32x32x2_f32 (etch_inter_so3conv32, ETCH_INTER_MFMA32=1;
    profiles/r03_inter_conv32_and_valu_overlap.txt) computes the same convolution; odd point counts leave its second point empty."""
    from etch_amd import _lib
    if not _lib.has_experiments():
        pytest.skip("opt-in experiment kernel: built only with ETCH_BUILD_EXPERIMENTS=1 (measured slower than the default path)")
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(cin + nn)
    b, p1, p2 = 2, 211, 101
    xyz = (torch.randn(b, 3, p1, generator=g) * 0.2).cuda()
    new_xyz = xyz[:, :, :p2].contiguous()
    ball = ops.ball_query(new_xyz, xyz, 0.25, nn)
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, 2, 0.25, 0.03, nn), 3).cuda()
    rk, W, Wp, bias = conv._derived()
    feats = torch.randn(b, p1, 60, cin, generator=g).cuda()
    ref, (m0, r0) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, want_stats=True)
    saved = ops.INTER_MFMA32
    ops.INTER_MFMA32 = True
    try:
        out, (m, r) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, want_stats=True, Wp32=conv._wp32(), order=ops.spatial_order(new_xyz))
        assert torch.equal(out, ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, Wp32=conv._wp32()))
    finally:
        ops.INTER_MFMA32 = saved
    assert float((out - ref).abs().max()) < 2e-6 * float(ref.abs().max())
    assert rel_err(m.cpu().numpy(), m0.cpu().numpy()) < 2e-6 and rel_err(r.cpu().numpy(), r0.cpu().numpy()) < 2e-6
