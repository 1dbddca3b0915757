"""Round-6 GPU tests: the persistent inter conv (csrc/so3conv_y.hip), operands of any scale through the operator API, the ADVICE r05
corner cases of the two-plane fp16 kernels, and the graph-replay pipeline."""
import numpy as np
import pytest
import torch

from etch_amd.utils.weights import load_seeded
from tests.test_gpu_encoder import _inter_conv_fp64

pytestmark = pytest.mark.gpu


def _inter_setup(cin, cout, nn, b, p1, p2, seed=3, radius=0.25, sigma=0.03):
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.randn(b, 3, p1, generator=g) * 0.2).cuda()
    new_xyz = xyz[:, :, :p2].contiguous()
    ball = ops.ball_query(new_xyz, xyz, radius, nn)
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, 1, radius, sigma, nn), 3).cuda()
    feats = torch.randn(b, p1, 60, cin, generator=g).cuda()
    return g, xyz, new_xyz, ball, conv, feats


@pytest.mark.parametrize("cin,cout,nn,p1,p2", [(32, 32, 32, 301, 149), (32, 64, 64, 211, 60), (64, 64, 32, 211, 101)])
@pytest.mark.parametrize("scale", [2.0 ** -10, 2.0 ** -5, 2.0 ** 5, 2.0 ** 12, 1e5, "channels"])
def test_inter_conv_features_of_any_scale(cin, cout, nn, p1, p2, scale):
    """VERDICT r05 item 4.  The operator API (vgtk so3conv/modules.py:92-128) has no domain restriction: features at 2^-10 .. 1e5 of unit scale, and with
    per-channel scales spanning four decades, through ops.inter_so3conv WITHOUT producer planes (they are then made per call, every scan times its own
    power of two, undone in the epilogue: etch_split2_planes_f16_scaled / `fsc`).  No inf / nan; as close to the fp64 formula as the fp32-MFMA kernel
    (exact for any scale) -- entitled-error rule: error <= 2 x the fp32 kernel's + 1e-7 of the output scale."""
    from etch_amd import ops
    g, xyz, new_xyz, ball, conv, feats = _inter_setup(cin, cout, nn, 2, p1, p2)
    if scale == "channels":
        feats = feats * (10.0 ** (torch.rand(cin, generator=g) * 4 - 2)).cuda()
    else:
        feats = feats * scale
    feats = feats.contiguous()
    rk, W, Wp, bias = conv._derived()
    bias = bias * 0                           # (a bias of unit size would hide the small-scale cases)
    new = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, order=ops.spatial_order(new_xyz), Wqh=conv._wqh(), kq=conv._kq())
    f32 = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma)
    assert bool(torch.isfinite(new).all())
    ref = _inter_conv_fp64(xyz, new_xyz, ball, feats, rk, W, bias, conv.sigma)
    s = float(ref.abs().max())
    e_new, e_f32 = float((new.double() - ref).abs().max()), float((f32.double() - ref).abs().max())
    assert e_f32 < 3e-6 * s and e_new <= 2.0 * e_f32 + 1e-7 * s, (e_new / s, e_f32 / s)


def test_inter_conv_scan_scaling_is_per_scan_and_the_model_planes_are_untouched():
    """The power of two of a scan comes from that scan alone: scan 0's result is bit for bit the same next to a neighbour 1e5 times larger and alone;
    planes that arrive WITH the features (the producer's, unit scale by construction) are gathered as they are."""
    from etch_amd import ops
    g, xyz, new_xyz, ball, conv, feats = _inter_setup(32, 64, 64, 3, 211, 60)
    feats[1] *= 1e5
    feats[2] *= 2.0 ** -9
    rk, W, Wp, bias = conv._derived()
    kw = dict(Wqh=conv._wqh(), kq=conv._kq())
    full = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, **kw)
    for i in range(3):
        solo = ops.inter_so3conv(xyz[i:i + 1].contiguous(), new_xyz[i:i + 1].contiguous(), ball[i:i + 1].contiguous(), feats[i:i + 1].contiguous(), rk, W, Wp, bias,
                                 conv.sigma, **kw)
        assert torch.equal(solo[0], full[i]), i
    planes, fsc = ops.split2_planes_f16_scaled(feats)
    k = torch.log2(fsc).cpu()
    assert torch.equal(k, k.round()) and float(k[1] - k[0]) >= 16 and float(k[2] - k[0]) == -9      # fsc = 2^-k: powers of two, per scan
    m = (feats * (1.0 / fsc).view(3, 1, 1, 1)).abs().amax((1, 2, 3)).cpu()
    assert bool(((m >= 8) & (m < 16)).all()), m
    assert torch.equal(planes[0, :, :, 0], (feats[0] / fsc[0]).half())


@pytest.mark.parametrize("cin,cout,nn,b,p2", [(32, 32, 32, 3, 5), (32, 64, 64, 1, 1), (64, 64, 32, 5, 700), (32, 32, 32, 16, 1250)])
def test_persistent_inter_conv_item_lists(cin, cout, nn, b, p2):
    """Round 6: persistent workgroups walk per-XCD item lists (two static items, then a work counter).  Fewer items than workgroups, one item, odd scan
    counts, many items per workgroup: every output row is written (the whole tensor equals the fp32-MFMA kernel's to 2e-6), ordered == plain order bit for
    bit, repeated launches bit for bit (whatever the dynamic distribution),."""
    from etch_amd import ops
    p1 = max(p2, 130)
    g, xyz, new_xyz, ball, conv, feats = _inter_setup(cin, cout, nn, b, p1, p2, radius=0.25 if p1 < 1000 else 0.113137, sigma=0.03 if p1 < 1000 else 0.0064)
    rk, W, Wp, bias = conv._derived()
    planes = ops.split2_planes_f16(feats)
    kw = dict(Wqh=conv._wqh(), kq=conv._kq(), feats_planes=planes)
    order = ops.spatial_order(new_xyz)
    f32 = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma)
    first = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, order=order, **kw)
    assert bool(torch.isfinite(first).all())
    assert float((first - f32).abs().max()) < 2e-6 * float(f32.abs().max())
    for rep in range(6):
        again, (m1, r1) = ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, conv.sigma, order=order if rep % 2 else None, want_stats=True, **kw)
        assert torch.equal(again, first), rep
    m0, r0 = ops.instnorm_stats(first)
    assert float((m1 - m0).abs().max()) < 1e-5 * float(m0.abs().max() + 1) and float((r1 / r0 - 1).abs().max()) < 1e-5


def test_attention_layer_per_matrix_weight_scaling_bound_at_s12():
    """the attention layers' weights are range-scaled per MATRIX; an output channel whose weight row lies 2^-s below the matrix maximum keeps a
    relative error of 2^(s - 29) (the planes' absolute floor 2^-25 against a row at 2^(4 - s)).  Pinned at s = 12: rows of the value
    projection at 2^-12 of the others -- their output channels must stay within 4 x 2^-17 of their own scale (+ the fp32 level), all other channels at the
    usual bar."""
    from etch_amd import ops
    from tests.test_gpu_heads import _mhsa_reference
    g = torch.Generator().manual_seed(12)
    T, s = 200, 12
    x = torch.randn(T, 60, 64, generator=g)
    wq, wk, wv, wc = (torch.randn(64, 64, generator=g) * 0.25 for _ in range(4))
    small = torch.tensor([3, 17, 40, 63])
    wv[small] *= 2.0 ** -s
    ref = _mhsa_reference(x, wq, wk, wv, wc, torch.zeros(64), 2).reshape(T * 60, 64)
    out = ops.mhsa_layer(x.cuda().view(T * 60, 64), wq.cuda(), wk.cuda(), wv.cuda(), mode=2).double().cpu()
    ch = ref.abs().amax(0)
    err = (out - ref).abs().amax(0) / ch
    big = torch.ones(64, dtype=torch.bool)
    big[small] = False
    assert float(err[big].max()) < 3e-6, float(err[big].max())
    assert float(err[small].max()) < 4 * 2.0 ** (s - 29) + 3e-6, (float(err[small].max()), 2.0 ** (s - 29))


def test_rows_and_tiles_at_the_bottom_of_the_float_range():
    """ADVICE r05.  (1) weight rows whose maximum is 1e-38 / subnormal / zero: the host's row powers are capped like the device's (ops._pow2_exp), the
    planes stay finite, and the conv's output for such a row is what the fp32 kernel gives (~0).  (2) a token tile at 1e-25: the scores' factor would
    underflow; it is held normal, the padded keys stay at probability 0 and the layer returns the uniform softmax the reference formula gives."""
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    from tests.test_gpu_heads import _mhsa_reference
    g = torch.Generator().manual_seed(4)
    conv = load_seeded(V.IntraSO3Conv(64, 64), 5).cuda()
    with torch.no_grad():
        conv.basic_conv.W[5] *= 1e-38 / float(conv.basic_conv.W[5].abs().max())
        conv.basic_conv.W[9] *= 1e-44 / float(conv.basic_conv.W[9].abs().max())
        conv.basic_conv.W[11] = 0.0
    Wp, bias, idx32, Wp32 = conv._derived()
    assert bool(torch.isfinite(conv._wqh.wsc).all()) and bool(torch.isfinite(conv._wqh.view(torch.float16).float()).all())
    x = torch.randn(2, 64, 60, 64, generator=g).cuda()
    mm, rr = ops.instnorm_stats(x)
    new = ops.intra_so3conv(x, idx32, Wp, bias * 0, 64, mm, rr, Wqh=conv._wqh)
    f32 = ops.intra_so3conv(x, idx32, Wp, bias * 0, 64, mm, rr)
    assert bool(torch.isfinite(new).all())
    assert float((new - f32).abs().max()) < 2e-6 * float(f32.abs().max())
    assert float(new[..., [5, 9, 11]].abs().max()) < 1e-30
    # (2)
    T = 40
    x = torch.randn(T, 60, 64, generator=g)
    x[::2] *= 1e-25
    ws = [torch.randn(64, 64, generator=g) * 0.2 for _ in range(4)]
    ref = _mhsa_reference(x, ws[0], ws[1], ws[2], ws[3], torch.zeros(64), 2).reshape(T, 60 * 64)
    out = ops.mhsa_layer(x.cuda().view(T * 60, 64), ws[0].cuda(), ws[1].cuda(), ws[2].cuda(), mode=2).double().cpu().reshape(T, 60 * 64)
    assert bool(torch.isfinite(out).all())
    e = (out - ref).abs().amax(1) / ref.abs().amax(1)
    assert float(e.max()) < 3e-6, float(e.max())


def test_intra_conv_without_statistics_takes_an_exact_form():
    """without (mean, rstd) the intra conv's rows are the caller's own, of unknown scale -- those calls do not take the two-plane fp16 form
    (values above 65 504 would become inf): rows at 1e6 and at 1e-6 come out finite and within the fp32 bar of the fp64 formula."""
    from etch_amd import ops
    from etch_amd import vgtk_so3conv as V
    g = torch.Generator().manual_seed(8)
    conv = load_seeded(V.IntraSO3Conv(32, 32), 5).cuda()
    Wp, bias, idx32, Wp32 = conv._derived()
    for scale in (1e6, 1e-6):
        x = (torch.randn(2, 50, 60, 32, generator=g) * scale).cuda()
        out = ops.intra_so3conv(x, idx32, Wp, bias * 0, 32, Wp32=Wp32, Wq=conv._wq, Wqh=conv._wqh)
        ref = torch.einsum("bpatc,oct->bpao", x.double()[:, :, conv.intra_idx.cuda()], conv.basic_conv.W.detach().double().view(32, 32, 12))
        assert bool(torch.isfinite(out).all())
        assert float((out.double() - ref).abs().max()) < 3e-6 * float(ref.abs().max()), scale


def test_graph_replay_pipeline_matches_the_synchronous_path(tmp_path):
    """etch_amd.pipeline.GraphPipeline (opt-in schedule: one HIP-graph replay per batch slot, two slots here): the same bits as one batch at a time."""
    from etch_amd.inference_demo import predict_smpl_batch
    from etch_amd.pipeline import GraphPipeline
    from tests.test_gpu_pipeline import make, scan
    args, model = make(tmp_path)
    B, N, nb = 2, 640, 4
    mk = lambda k: torch.from_numpy(np.stack([scan(300 + 10 * k + b, N) for b in range(B)])).cuda()
    ref = [predict_smpl_batch(args, model, mk(k), "neutral") for k in range(nb)]
    pipe = GraphPipeline(args, model, B, N, "neutral", max_in_flight=2)
    got = list(pipe.run(mk(k) for k in range(nb)))
    same = lambda a, b: np.array_equal(np.asarray(a.cpu() if torch.is_tensor(a) else a), np.asarray(b.cpu() if torch.is_tensor(b) else b), equal_nan=True)
    for (m0, mk0, v0, i0), (m1, mk1, v1, i1) in zip(ref, got):
        assert same(mk0, mk1) and same(v0, v1)
        for a, b in zip(i0, i1):
            assert same(a, b)
        assert same(m0[0].vertices, m1[0].vertices)


@pytest.mark.parametrize("b,n,stride", [(3, 5000, 2), (2, 20000, 2), (4, 1024, 2), (1, 777, 3)])
def test_fps_pair_equals_the_two_separate_samplings(b, n, stride):
    """etch_fps_pair: the encoder's FPS (grouping_cuda_kernel.cu:352-466) and the nets' first FPS level (sampling_cuda_kernel.cu:15-129) of the same
    scans in one launch of 2 b workgroups -- bit for bit the picks of the separate launches, on random scans, on a lattice full of ties and with points at
    the origin (which only the vgtk flavour skips)."""
    from etch_amd import ops
    from etch_amd.models import pointops
    from etch_amd.models.pointtransformer_seg import downsampled_offsets
    g = torch.Generator().manual_seed(n + b)
    for kind in ("random", "lattice"):
        if kind == "random":
            pts = torch.randn(b, n, 3, generator=g) * torch.tensor([0.14, 0.31, 0.085])
        else:
            pts = torch.randint(0, 7, (b, n, 3), generator=g).float() * 0.05
        pts[:, 5] = 0.0                                   # a point at the origin
        pts = pts.cuda().contiguous()
        xyz = pts.permute(0, 2, 1).contiguous()
        m = -(-n // stride)
        oh = [n * (i + 1) for i in range(b)]
        o = pointops.offsets_tensor(oh, pts.device)
        n_o = downsampled_offsets(oh, 4)
        n_o_t = pointops.offsets_tensor(n_o, pts.device)
        ia, ib = ops.fps_pair(xyz, m, pts.view(-1, 3), o, n_o_t, n_o)
        ra = ops.furthest_point_sampling(xyz, m, split=1)
        rb = ops.furthestsampling(pts.view(-1, 3), o, n_o_t, oh, n_o, split=1)
        assert torch.equal(ia, ra) and torch.equal(ib, rb), kind


def test_model_with_and_without_the_paired_fps_is_bitwise_identical(tmp_path):
    from tests.test_gpu_pipeline import make, scan
    args, model = make(tmp_path)
    pts = torch.from_numpy(np.stack([scan(500 + b, 2048) for b in range(2)])).cuda()
    items = ["confidence", "direction", "magnitude"]
    out = {}
    for on in (True, False, True):
        model.fps_pair = on
        with torch.no_grad():
            res, _ = model(pts, items, "standard_vector")
        torch.cuda.synchronize()
        if not out:
            out = {k: v.clone() for k, v in res.items()}
        for k in out:
            assert torch.equal(res[k], out[k]), (k, on)
    from etch_amd import ops
    assert not ops._FPS_READY


# ---------------------------------------------------------------------------------------------- one-launch reductions of the training kernels
@pytest.mark.gpu
@pytest.mark.parametrize("R,M,N", [(80000, 32, 32), (5000, 64, 1536), (300000, 64, 128), (2000, 4, 3), (50, 512, 512), (777, 32, 48), (1, 16, 16),
                                   (4097, 33, 31), (1250, 512, 64)])
def test_gemm_tn_one_launch_is_bitwise_the_two_launch_form_and_exact_to_fp64(R, M, N):
    """etch_gemm_tn_fused (the workgroup of a tile that finishes last sums the tile's partials in split order) against etch_gemm_tn (partials + a
    reduction launch): the same bits, with and without accumulation; both within fp32 rounding of the fp64 product (fp64 matrix cores); the counters
    are left at zero; strided operands (column windows of wider matrices)."""
    import ctypes
    from etch_amd import _lib, autograd as A
    from etch_amd.ops import _ptr, _stream
    rng = np.random.default_rng(R + M + N)
    a_full = torch.from_numpy(rng.standard_normal((R, M + 5)).astype(np.float32)).cuda()
    b_full = torch.from_numpy(rng.standard_normal((R, N + 3)).astype(np.float32)).cuda()
    a, bm = a_full[:, 2:2 + M], b_full[:, 1:1 + N]
    ref = (a.double().t() @ bm.double())
    c0 = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).cuda()
    lib = _lib.lib()
    nws = lib.etch_gemm_tn_workspace_floats(ctypes.c_long(R), M, N)
    for accumulate in (0, 1):
        two, one = c0.clone(), c0.clone()
        ws = torch.empty((nws,), dtype=torch.float32, device="cuda")
        _lib.check(lib.etch_gemm_tn(ctypes.c_long(R), M, N, _ptr(a), ctypes.c_long(a.stride(0)), _ptr(bm), ctypes.c_long(bm.stride(0)), _ptr(two), accumulate,
                                    _ptr(ws), _stream()), "etch_gemm_tn")
        ws2 = torch.empty((nws,), dtype=torch.float32, device="cuda")
        counters = A.reduce_counters(a.device)
        _lib.check(lib.etch_gemm_tn_fused(ctypes.c_long(R), M, N, _ptr(a), ctypes.c_long(a.stride(0)), _ptr(bm), ctypes.c_long(bm.stride(0)), _ptr(one),
                                          accumulate, _ptr(ws2), _ptr(counters), _stream()), "etch_gemm_tn_fused")
        assert torch.equal(one, two)
        want = (ref + c0.double()) if accumulate else ref
        assert float((one.double() - want).abs().max()) <= 1.2e-7 * float(want.abs().max()) + 1e-30
        assert int(counters.abs().sum()) == 0
    again = A.gemm_tn(a, bm)
    assert torch.equal(again, A.gemm_tn(a, bm))                     # run to run


@pytest.mark.gpu
@pytest.mark.parametrize("R,C", [(80000, 32), (5000, 131), (3, 512), (1, 1), (300000, 64), (1000, 4100)])
def test_colsum_one_launch(R, C):
    """etch_colsum_fused: fp64 column sums in a fixed order, one launch (C > 4096: the two-launch form behind the same entry point)."""
    from etch_amd import autograd as A
    rng = np.random.default_rng(R + C)
    x = torch.from_numpy((rng.standard_normal((R, C)) * 3 + 0.7).astype(np.float32)).cuda()
    s = A.colsum(x)
    ref = x.double().sum(0)
    assert float((s.double() - ref).abs().max()) <= 1.2e-7 * float(ref.abs().max()) + 1e-30
    assert torch.equal(s, A.colsum(x))
    assert int(A.reduce_counters(x.device).abs().sum()) == 0


@pytest.mark.gpu
def test_batch_norm_train_forward_in_two_launches_matches_torch_over_many_calls():
    """etch_bn_train_forward / etch_bn_backward_fused behind autograd_pt.batch_norm: ten consecutive train() calls on one module against
    torch.nn.BatchNorm1d (+ ReLU) -- outputs, running statistics after every call (momentum 0.1 and the cumulative average momentum=None),
    num_batches_tracked, gradients -- and the counters are zero afterwards."""
    from etch_amd import autograd_pt as P, autograd as A
    rng = np.random.default_rng(5)
    for C, momentum in ((96, 0.1), (7, None), (512, 0.3)):
        m, m2 = torch.nn.BatchNorm1d(C, momentum=momentum).cuda().train(), torch.nn.BatchNorm1d(C, momentum=momentum).cuda().train()
        with torch.no_grad():
            w = torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).cuda()
            m.weight.copy_(w), m2.weight.copy_(w)
        for it in range(10):
            R = int(rng.integers(2, 3000))
            x = torch.from_numpy((rng.standard_normal((R, C)) * (1 + it) + it).astype(np.float32)).cuda().requires_grad_()
            x2 = x.detach().clone().requires_grad_()
            y, y2 = P.batch_norm(x, m, relu=True), torch.relu(m2(x2))
            assert float((y - y2).detach().abs().max()) <= 2e-6 * float(y2.detach().abs().max())
            for got, want in ((m.running_mean, m2.running_mean), (m.running_var, m2.running_var)):
                assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
            assert int(m.num_batches_tracked) == int(m2.num_batches_tracked) == it + 1
            g = torch.randn_like(y)
            y.backward(g), y2.backward(g)
            assert float((x.grad - x2.grad).abs().max()) <= 2e-4 * float(x2.grad.abs().max()) + 1e-6
        assert float((m.weight.grad - m2.weight.grad).abs().max()) <= 1e-4 * float(m2.weight.grad.abs().max())
        assert float((m.bias.grad - m2.bias.grad).abs().max()) <= 1e-4 * float(m2.bias.grad.abs().max())
    assert int(A.reduce_counters(torch.device("cuda", 0)).abs().sum()) == 0


@pytest.mark.gpu
def test_last_workgroup_reductions_soak_under_contention():
    """The hand-over of the one-launch reductions (common.h: etch_last_block -- partials, release fence, one atomic per workgroup, the last arrival
    acquires and sums) across the eight XCDs' L2s: 400 rounds of randomly shaped weight gradients (bitwise against the two-launch form), column sums
    and BatchNorm statistics (bitwise run to run) while a second stream keeps the memory system busy.  A lost or stale partial shows as a mismatch."""
    import ctypes
    from etch_amd import _lib, autograd as A, autograd_pt as P
    from etch_amd.ops import _ptr, _stream
    lib = _lib.lib()
    rng = np.random.default_rng(12)
    big = torch.randn(64 * 1024 * 1024 // 4, device="cuda")
    side = torch.cuda.Stream()
    shapes = [(80000, 32, 32), (20000, 128, 16), (5000, 64, 64), (1250, 128, 128), (300000, 64, 8), (4992, 256, 4), (312, 256, 256), (40000, 8, 64)]
    data = {}
    for (R, M, N) in shapes:
        data[(R, M, N)] = (torch.randn(R, M, device="cuda"), torch.randn(R, N, device="cuda"))
    ref = {}
    for it in range(400):
        if it % 8 == 0:
            with torch.cuda.stream(side):
                for _ in range(4):
                    big.mul_(1.0000001)
        R, M, N = shapes[int(rng.integers(len(shapes)))]
        a, bm = data[(R, M, N)]
        one = A.gemm_tn(a, bm)
        cs = A.colsum(a)
        m = torch.nn.BatchNorm1d(M).cuda().train()
        y = P.batch_norm(a, m, relu=True)
        key = (R, M, N)
        if key not in ref:
            two = torch.empty_like(one)
            ws = torch.empty((lib.etch_gemm_tn_workspace_floats(ctypes.c_long(R), M, N),), dtype=torch.float32, device="cuda")
            _lib.check(lib.etch_gemm_tn(ctypes.c_long(R), M, N, _ptr(a), ctypes.c_long(a.stride(0)), _ptr(bm), ctypes.c_long(bm.stride(0)), _ptr(two), 0, _ptr(ws),
                                        _stream()), "etch_gemm_tn")
            ref[key] = (two, cs.clone(), y.detach().clone(), m.running_var.clone())
        two, cs0, y0, rv0 = ref[key]
        assert torch.equal(one, two), (it, key)
        assert torch.equal(cs, cs0), (it, key)
        assert torch.equal(y.detach(), y0) and torch.equal(m.running_var, rv0), (it, key)
    torch.cuda.synchronize()
    assert int(A.reduce_counters(torch.device("cuda", 0)).abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,nn,b,p1,p2", [(32, 32, 32, 2, 301, 149), (32, 64, 64, 1, 400, 200), (64, 64, 32, 2, 211, 101), (16, 32, 9, 1, 100, 37)])
def test_inter_conv_feature_gradient_from_the_target_side(cin, cout, nn, b, p1, p2):
    """etch_inter_dfeat_slots + the ordered segment sum (every dX1 tile read once) against inter_dfeat_kernel (every slot re-reads its target's tile):
    the same gradient to fp32 rounding (the slots' contributions are rounded before they are summed instead of inside one running sum), in one chunk
    and in several, bitwise run to run; the weight / bias gradients do not depend on the path."""
    from etch_amd import autograd as A
    g, xyz, new_xyz, ball, conv, feats = _inter_setup(cin, cout, nn, b, p1, p2, seed=cin + nn)
    rk = conv.rotated_kernels()
    G = torch.randn(b, p2, 60, cout, generator=g).cuda()

    def grads(slot, chunk_bytes):
        old = A.SLOT_DFEAT, A.CHUNK_BYTES
        A.SLOT_DFEAT, A.CHUNK_BYTES = slot, chunk_bytes
        try:
            f = feats.detach().clone().requires_grad_()
            W = conv.basic_conv.W.detach().clone().requires_grad_()
            bs = conv.basic_conv.bias.detach().clone().requires_grad_()
            y = A.inter_so3conv(f, W, bs, xyz, new_xyz, ball, rk, conv.sigma, chunk=40)
            (y * G).sum().backward()
            return f.grad, W.grad, bs.grad
        finally:
            A.SLOT_DFEAT, A.CHUNK_BYTES = old
    f0, W0, b0 = grads(False, 1.2e9)
    f1, W1, b1 = grads(True, 1.2e9)            # one chunk
    f2, W2, b2 = grads(True, 1.0)              # chunks of 40 points
    f3, _, _ = grads(True, 1.0)
    s = float(f0.abs().max())
    assert float((f1 - f0).abs().max()) <= 3e-6 * s and float((f2 - f0).abs().max()) <= 3e-6 * s
    assert torch.equal(f2, f3) and torch.equal(W0, W1) and torch.equal(b0, b1)
    assert float((W2 - W0).abs().max()) <= 1e-6 * float(W0.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("b,rows,C,planes", [(2, 1000, 32, "f16"), (3, 777, 64, False), (1, 60, 32, "f16")])
def test_one_channel_skip_branch_folded_into_the_final_pass(b, rows, C, planes):
    """etch_instnorm_act_add_k1_planes_f16: lrelu(IN(x1)) + lrelu(IN(w f + bias)) with the one-channel conv and its InstanceNorm as a slope / offset per
    (scan, channel), against the un-folded chain (etch_linear -> etch_instnorm_stats -> etch_instnorm_act_add) and the fp64 formula of so3conv.py:171-183;
    random features (the model feeds ones: see the model-level test below), constant features (variance 0: the branch is exactly 0)."""
    from etch_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x1 = torch.randn(b, rows, C, generator=g).cuda()
    m1, r1 = ops.instnorm_stats(x1)
    w = (torch.randn(C, 1, generator=g) * 0.7).cuda()
    bias = (torch.randn(C, generator=g) * 0.3).cuda()
    for kind in ("random", "ones"):
        f = (torch.randn(b, rows, generator=g) * 1.3 + 0.4).cuda() if kind == "random" else torch.ones(b, rows).cuda()
        var, mean = torch.var_mean(f.double(), dim=1, unbiased=False, keepdim=True)
        w64, b64 = w.double().view(1, -1), bias.double().view(1, -1)
        rstd = torch.rsqrt(w64 * w64 * var + 1e-5)
        res = ops.instnorm_act_add_k1(x1, m1, r1, f, (w64 * rstd).float(), (-(w64 * mean) * rstd).float(), want_planes=planes)
        out, pl = res if planes else (res, None)
        s = ops.linear(f.view(-1, 1), w, bias=bias).view(b, rows, C)
        m3, r3 = ops.instnorm_stats(s)
        res0 = ops.instnorm_act_add(x1, m1, r1, s, m3, r3, want_planes=planes)
        out0, pl0 = res0 if planes else (res0, None)
        lr = torch.nn.functional.leaky_relu
        s64 = f.double().unsqueeze(-1) * w.double().view(1, 1, -1) + bias.double()
        ref = lr((x1.double() - x1.double().mean(1, keepdim=True)) / torch.sqrt(x1.double().var(1, unbiased=False, keepdim=True) + 1e-5), 0.01) + \
            lr((s64 - s64.mean(1, keepdim=True)) / torch.sqrt(s64.var(1, unbiased=False, keepdim=True) + 1e-5), 0.01)
        sc = float(ref.abs().max())
        e_new, e_old = float((out.double() - ref).abs().max()), float((out0.double() - ref).abs().max())
        assert e_new <= 2.0 * e_old + 2e-6 * sc, (kind, e_new / sc, e_old / sc)
        if pl is not None:
            back = pl.float().sum(-2)                       # h + l
            assert float((back - out).abs().max()) <= 2e-6 * sc


@pytest.mark.gpu
def test_model_with_and_without_the_folded_skip_branch(tmp_path):
    """GT_network_equiv with ETCH_SKIP_K1_FOLD on / off: the same outputs to fp32 rounding of the first block's skip branch."""
    import types
    from etch_amd import constants as K
    from etch_amd.models.models_pointcloud import GT_network_equiv
    from etch_amd.models.so3conv import SeparableSO3ConvBlock
    dev = torch.device("cuda", 0)
    args = types.SimpleNamespace(output_folder=str(tmp_path), EPN_input_radius=0.4, EPN_layer_num=2, device=dev, markerset=K.default_markerset(), scale_magnitude=10)
    model = load_seeded(GT_network_equiv(option=args), 1).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    pts = (torch.randn(2, 1500, 3, generator=g) * torch.tensor([0.14, 0.31, 0.085])).cuda()
    items = ["confidence", "direction", "magnitude"]
    old = SeparableSO3ConvBlock.fold_k1_skip
    try:
        res = {}
        for flag in (True, False):
            SeparableSO3ConvBlock.fold_k1_skip = flag
            with torch.no_grad():
                res[flag] = model(pts, items, "standard_vector")[0]
    finally:
        SeparableSO3ConvBlock.fold_k1_skip = old
    for k in ("magnitude", "confidences", "part_labels"):
        a, b_ = res[True][k], res[False][k]
        assert float((a - b_).abs().max()) <= 2e-5 * float(b_.abs().max()) + 1e-6, k


@pytest.mark.gpu
def test_first_conv_with_all_ones_features_passed_as_null():
    """cin = 1 with feats == NULL (all-ones occupancy features: no gather, no product) is bitwise the kernel fed a tensor of ones, statistics included."""
    from etch_amd import ops
    from etch_amd.models.so3conv import _occupancy_ones
    g, xyz, new_xyz, ball, conv, _ = _inter_setup(1, 32, 64, 2, 400, 200, seed=9)
    rk, W, Wp, bias = conv._derived()
    ones = torch.ones(2, 400, 60, 1, device="cuda")
    tagged = _occupancy_ones(2, 400, 60, ones.device)
    assert getattr(tagged, "_etch_constant", None) == 1.0 and bool((tagged == 1).all())
    a, (ma, ra) = ops.inter_so3conv(xyz, new_xyz, ball, ones, rk, W, Wp, bias, conv.sigma, want_stats=True)
    c, (mc, rc) = ops.inter_so3conv(xyz, new_xyz, ball, tagged, rk, W, Wp, bias, conv.sigma, want_stats=True)
    assert torch.equal(a, c) and torch.equal(ma, mc) and torch.equal(ra, rc)
