"""GPU tests of the backward kernels: gradients of the fused inter / intra SO(3) convolutions and of
InstanceNorm + LeakyReLU against torch.autograd through the oracle's un-fused restatement of the reference forms
(vgtk/so3conv/functional.py:224-378, modules.py:33-39; so3conv.py:36-44) evaluated in fp64; reproducibility run to run."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _parity import rel_err

pytestmark = pytest.mark.gpu


def scan(seed, n):
    return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)


@pytest.mark.parametrize("stride,lazy,nn,cin,cout,chunk", [(2, False, 16, 16, 32, 64), (1, True, 32, 32, 32, 100), (2, False, 32, 64, 64, 256)])
def test_inter_conv_gradients_vs_autograd_of_the_unfused_form(stride, lazy, nn, cin, cout, chunk):
    import etch_amd.vgtk_so3conv as V
    from etch_amd import autograd as A
    from etch_amd.utils.weights import load_seeded
    from oracle import stage1 as S1
    b, n, radius, sigma = 2, 300, 0.2, 0.02
    rng = np.random.default_rng(1)
    xyz = torch.from_numpy(np.stack([scan(30 + i, n).T.copy() for i in range(b)]))
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, stride, radius, sigma, nn, lazy_sample=lazy), 4).cuda().eval()
    g_ref, ball, sidx, new_xyz = S1.inter_grouping(xyz, stride, radius, nn, lazy)
    p2 = new_xyz.shape[2]
    # reference gradients: fp64 autograd through weights -> grouping einsum -> GEMM
    feats = torch.from_numpy(rng.standard_normal((b, cin, n, 60))).double().requires_grad_()
    W = conv.basic_conv.W.detach().cpu().double().requires_grad_()
    bias = conv.basic_conv.bias.detach().cpu().double().requires_grad_()
    w = S1.inter_weights(g_ref.double(), conv.anchors.cpu().double(), conv.kernels.cpu().double(), sigma)
    nf = S1.inter_feat_grouping(ball, w, torch.cat((feats, torch.zeros(b, cin, 1, 60, dtype=torch.float64)), 2))
    y_ref = S1.basic_so3conv(W, bias, nf)                                   # (b, cout, p2, 60)
    G = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)))
    (y_ref * G).sum().backward()
    # ours
    rk, Wd, Wp, bd = conv._derived()
    f_cl = feats.detach().float().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    Wg = conv.basic_conv.W.detach().clone().requires_grad_()
    bg = conv.basic_conv.bias.detach().clone().requires_grad_()
    y = A.inter_so3conv(f_cl, Wg, bg, xyz.cuda(), new_xyz.cuda(), ball.cuda(), rk, sigma, chunk=chunk)
    assert rel_err(y.detach().cpu().permute(0, 3, 1, 2).numpy(), y_ref.detach().numpy()) < 1e-4
    (y * G.float().permute(0, 2, 3, 1).contiguous().cuda()).sum().backward()
    assert rel_err(Wg.grad.cpu().numpy(), W.grad.numpy()) < 1e-4
    assert rel_err(bg.grad.cpu().numpy(), bias.grad.numpy()) < 1e-4
    assert rel_err(f_cl.grad.cpu().permute(0, 3, 1, 2).numpy(), feats.grad.numpy()) < 1e-4
    # reproducible: a second backward gives the same bits
    f2 = f_cl.detach().clone().requires_grad_()
    W2 = Wg.detach().clone().requires_grad_()
    b2 = bg.detach().clone().requires_grad_()
    y2 = A.inter_so3conv(f2, W2, b2, xyz.cuda(), new_xyz.cuda(), ball.cuda(), rk, sigma, chunk=chunk)
    (y2 * G.float().permute(0, 2, 3, 1).contiguous().cuda()).sum().backward()
    assert torch.equal(W2.grad, Wg.grad) and torch.equal(f2.grad, f_cl.grad) and torch.equal(b2.grad, bg.grad)


def test_first_conv_weight_gradient_single_input_channel():
    """cin = 1 (the encoder's first conv: constant occupancy features, only W and bias receive gradients)."""
    import etch_amd.vgtk_so3conv as V
    from etch_amd import autograd as A
    from etch_amd.utils.weights import load_seeded
    from oracle import stage1 as S1
    b, n, nn, cout, radius, sigma = 1, 260, 24, 32, 0.2, 0.02
    xyz = torch.from_numpy(np.stack([scan(40, n).T.copy()]))
    conv = load_seeded(V.InterSO3Conv(1, cout, 1, 1, radius, sigma, nn, lazy_sample=True), 6).cuda().eval()
    g_ref, ball, sidx, new_xyz = S1.inter_grouping(xyz, 1, radius, nn, True)
    feats = torch.ones(b, 1, n, 60, dtype=torch.float64)
    W = conv.basic_conv.W.detach().cpu().double().requires_grad_()
    bias = conv.basic_conv.bias.detach().cpu().double().requires_grad_()
    w = S1.inter_weights(g_ref.double(), conv.anchors.cpu().double(), conv.kernels.cpu().double(), sigma)
    y_ref = S1.basic_so3conv(W, bias, S1.inter_feat_grouping(ball, w, torch.cat((feats, torch.zeros(b, 1, 1, 60, dtype=torch.float64)), 2)))
    G = torch.from_numpy(np.random.default_rng(2).standard_normal(tuple(y_ref.shape)))
    (y_ref * G).sum().backward()
    rk = conv._derived()[0]
    Wg = conv.basic_conv.W.detach().clone().requires_grad_()
    bg = conv.basic_conv.bias.detach().clone().requires_grad_()
    y = A.inter_so3conv(torch.ones(b, n, 60, 1).cuda(), Wg, bg, xyz.cuda(), new_xyz.cuda(), ball.cuda(), rk, sigma, chunk=128)
    (y * G.float().permute(0, 2, 3, 1).contiguous().cuda()).sum().backward()
    assert rel_err(Wg.grad.cpu().numpy(), W.grad.numpy()) < 1e-4 and rel_err(bg.grad.cpu().numpy(), bias.grad.numpy()) < 1e-4


@pytest.mark.parametrize("C", [16, 32, 64])
def test_intra_conv_gradients_vs_autograd(C):
    import etch_amd.vgtk_so3conv as V
    from etch_amd import autograd as A
    from etch_amd.utils.weights import load_seeded
    from oracle import stage1 as S1
    b, p = 2, 37
    rng = np.random.default_rng(C)
    conv = load_seeded(V.IntraSO3Conv(C, C), 7).cuda().eval()
    ii = conv.intra_idx.cpu()
    x = torch.from_numpy(rng.standard_normal((b, C, p, 60))).double().requires_grad_()
    W = conv.basic_conv.W.detach().cpu().double().requires_grad_()
    bias = conv.basic_conv.bias.detach().cpu().double().requires_grad_()
    gf = x.index_select(3, ii.view(-1)).view(b, C, p, 60, 12).permute(0, 1, 4, 2, 3).contiguous()        # functional.py:343-344
    y_ref = S1.basic_so3conv(W, bias, gf)
    G = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)))
    (y_ref * G).sum().backward()
    x_cl = x.detach().float().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    Wg = conv.basic_conv.W.detach().clone().requires_grad_()
    bg = conv.basic_conv.bias.detach().clone().requires_grad_()
    y = A.intra_so3conv(x_cl, Wg, bg, conv.intra_idx)
    assert rel_err(y.detach().cpu().permute(0, 3, 1, 2).numpy(), y_ref.detach().numpy()) < 1e-4
    (y * G.float().permute(0, 2, 3, 1).contiguous().cuda()).sum().backward()
    assert rel_err(x_cl.grad.cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy()) < 1e-4
    assert rel_err(Wg.grad.cpu().numpy(), W.grad.numpy()) < 1e-4
    assert rel_err(bg.grad.cpu().numpy(), bias.grad.numpy()) < 1e-4


@pytest.mark.parametrize("C,p", [(32, 41), (64, 130)])
def test_instancenorm_leaky_relu_gradient_vs_autograd(C, p):
    from etch_amd import autograd as A
    rng = np.random.default_rng(C + p)
    x = torch.from_numpy(rng.standard_normal((2, C, p, 60)) * 1.7 + 0.3).double().requires_grad_()
    y_ref = F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.01)
    G = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)))
    (y_ref * G).sum().backward()
    x_cl = x.detach().float().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
    y = A.instnorm_leaky_relu(x_cl)
    assert rel_err(y.detach().cpu().permute(0, 3, 1, 2).numpy(), y_ref.detach().numpy()) < 1e-5
    (y * G.float().permute(0, 2, 3, 1).contiguous().cuda()).sum().backward()
    assert rel_err(x_cl.grad.cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy()) < 1e-4


def test_gemm_tn_and_colsum():
    from etch_amd import autograd as A
    rng = np.random.default_rng(0)
    for R, M, N in ((5000, 64, 1536), (777, 32, 48), (33, 16, 16)):
        a = torch.from_numpy(rng.standard_normal((R, M)).astype(np.float32)).cuda()
        bm = torch.from_numpy(rng.standard_normal((R, N)).astype(np.float32)).cuda()
        c = A.gemm_tn(a, bm)
        ref = a.double().cpu().t() @ bm.double().cpu()
        assert rel_err(c.cpu().numpy(), ref.numpy()) < 1e-5
        c2 = A.gemm_tn(a, bm, out=c.clone(), accumulate=True)
        assert rel_err(c2.cpu().numpy(), 2 * ref.numpy()) < 1e-5
        assert rel_err(A.colsum(a).cpu().numpy(), a.double().cpu().sum(0).numpy()) < 1e-6


@pytest.mark.parametrize("T,residual", [(1, True), (37, True), (37, False)])
def test_mhsa_layer_gradients_vs_autograd_of_the_oracle_form(T, residual):
    """etch_amd.autograd.mhsa_layer (fused forward; attention-core backward kernel + matrix-core GEMMs) against fp64 autograd through
    the oracle's restatement of MultiHeadAttention (direction_backbones.py:132-194); reproducible run to run."""
    from etch_amd import autograd as A
    from oracle import stage1 as S1
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, 60, 64, generator=g)
    W = [torch.randn(64, 64, generator=g) * 0.2 for _ in range(4)]
    bc = torch.randn(64, generator=g) * 0.1
    G = torch.randn(T, 60, 64, generator=g)
    sd = {"l.query_transform.weight": W[0], "l.key_transform.weight": W[1], "l.value_transform.weight": W[2], "l.head_combine.weight": W[3],
          "l.head_combine.bias": bc}
    sd64 = {k: v.double().requires_grad_() for k, v in sd.items()}
    x64 = x.double().requires_grad_()
    y_ref = S1.mhsa_layer(sd64, "l.", x64)
    if residual:
        y_ref = y_ref + x64
    (y_ref * G.double()).sum().backward()
    ref = [x64.grad] + [sd64[k].grad for k in sd]

    def ours():
        ts = [t.cuda().requires_grad_() for t in [x] + W + [bc]]
        y = A.mhsa_layer(*ts, residual=residual)
        (y * G.cuda()).sum().backward()
        return y.detach().cpu(), [t.grad.cpu() for t in ts]
    y, grads = ours()
    assert rel_err(y.numpy(), y_ref.detach().numpy()) < 1e-5
    for name, a, b in zip(["x", "wq", "wk", "wv", "wc", "bc"], grads, ref):
        assert rel_err(a.numpy(), b.numpy()) < 1e-4, name
    _, again = ours()
    for a, b in zip(grads, again):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,c,ns", [(50, 32, 8), (300, 64, 16), (40, 256, 16)])
def test_pt_vector_attention_gradients_vs_fp64_autograd(n, c, ns):
    """etch_amd.autograd.pt_vector_attention against fp64 autograd through the oracle's statement of PointTransformerLayer's core
    (oracle/stage1.py::pt_layer after the q/k/v Linear layers; eval BatchNorm as scale / shift); reproducible run to run."""
    from etch_amd import autograd as A
    cs = c // 8
    g = torch.Generator().manual_seed(n + c)
    rn = lambda *s: torch.randn(*s, generator=g)
    p = rn(n, 3) * 0.2
    xq, xk, xv = rn(n, c), rn(n, c), rn(n, c)
    d2 = ((p[:, None] - p[None]) ** 2).sum(-1)
    idx = d2.topk(ns, largest=False).indices.int()
    P = dict(W0=rn(3, 3) * 2, b0=rn(3) * 0.1, s_p=rn(3).abs() + 0.5, t_p=rn(3) * 0.1, W3=rn(c, 3), b3=rn(c) * 0.1, s_w0=rn(c).abs() + 0.5, t_w0=rn(c) * 0.1,
             W2=rn(cs, c) / c ** 0.5, b2=rn(cs) * 0.1, s_w3=rn(cs).abs() + 0.5, t_w3=rn(cs) * 0.1, W5=rn(cs, cs) / cs ** 0.5, b5=rn(cs) * 0.1)
    order = ["W0", "b0", "s_p", "t_p", "W3", "b3", "s_w0", "t_w0", "W2", "b2", "s_w3", "t_w3", "W5", "b5"]
    diff = ["xq", "xk", "xv", "W0", "b0", "W3", "b3", "W2", "b2", "W5", "b5"]
    Gd = rn(n, c)

    def ref(xq, xk, xv, P):
        li = idx.long()
        r = p.double()[li] - p.double()[:, None]                                  # (n, ns, 3)
        h = F.relu(F.linear(r, P["W0"], P["b0"]) * P["s_p"] + P["t_p"])
        pr = F.linear(h, P["W3"], P["b3"])
        w = F.relu((xk[li] - xq[:, None] + pr) * P["s_w0"] + P["t_w0"])
        w = F.relu(F.linear(w, P["W2"], P["b2"]) * P["s_w3"] + P["t_w3"])
        w = F.softmax(F.linear(w, P["W5"], P["b5"]), dim=1)
        return ((xv[li] + pr).view(n, ns, 8, cs) * w.unsqueeze(2)).sum(1).view(n, c)
    t64 = {k: v.double().requires_grad_(k in diff) for k, v in dict(xq=xq, xk=xk, xv=xv, **P).items()}
    y_ref = ref(t64["xq"], t64["xk"], t64["xv"], t64)
    (y_ref * Gd.double()).sum().backward()

    def ours():
        t = {k: v.cuda().requires_grad_(k in diff) for k, v in dict(xq=xq, xk=xk, xv=xv, **P).items()}
        y = A.pt_vector_attention(p.cuda(), t["xq"], t["xk"], t["xv"], idx.cuda(), *[t[k] for k in order])
        (y * Gd.cuda()).sum().backward()
        return y.detach().cpu(), {k: t[k].grad.cpu() for k in diff}
    y, grads = ours()
    assert rel_err(y.numpy(), y_ref.detach().numpy()) < 1e-5
    for k in diff:
        if k == "b5":      # softmax over the neighbours is shift-invariant: the gradient is analytically zero (1e-15 in fp64)
            assert np.abs(grads[k].numpy()).max() < 1e-5 * np.abs(t64["W5"].grad.numpy()).max()
        else:
            assert rel_err(grads[k].numpy(), t64[k].grad.numpy()) < 1e-4, k
    _, again = ours()
    for k in diff:
        assert torch.equal(grads[k], again[k]), k


def test_so3_mean_dir_backward_vs_autograd_of_the_svd_form():
    """etch_so3_mean_dir_backward (derivative of the polar factor through the 3x3 Sylvester equation) against fp64 autograd through the
    oracle's torch.svd restatement of so3conv.py:186-225, on well-conditioned weight sets incl. reflections (det(U V^T) = -1)."""
    from etch_amd import autograd as A
    from etch_amd import constants as K
    from oracle import stage1 as S1
    rng = np.random.default_rng(3)
    T = 300
    w = rng.uniform(-0.2, 0.2, (T, 60))
    w[np.arange(T), rng.integers(0, 60, T)] += 3.0
    w[::7] *= -1.0                                                   # Ce -> -Ce: the nearest rotation needs the det fix
    anchors = torch.from_numpy(K.get_anchors())
    gd = rng.standard_normal((T, 3))
    w64 = torch.from_numpy(w).requires_grad_()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        R, Ce, sv = S1.so3_mean(anchors.double(), w64)
    finally:
        torch.set_default_dtype(old)
    assert float(torch.det(Ce.detach()).min()) < 0 < float(torch.det(Ce.detach()).max())
    (R[:, :, 2] * torch.from_numpy(gd)).sum().backward()
    wg = torch.from_numpy(w.astype(np.float32)).cuda().requires_grad_()
    d = A.so3_mean_dir(wg, anchors.cuda())
    assert np.abs(d.detach().cpu().numpy() - R[:, :, 2].detach().numpy()).max() < 1e-5
    (d * torch.from_numpy(gd.astype(np.float32)).cuda()).sum().backward()
    assert rel_err(wg.grad.cpu().numpy(), w64.grad.numpy()) < 1e-4
    g1 = wg.grad.clone()
    wg.grad = None
    (A.so3_mean_dir(wg, anchors.cuda()) * torch.from_numpy(gd.astype(np.float32)).cuda()).sum().backward()
    assert torch.equal(g1, wg.grad)


def test_propagation_backward_vs_autograd():
    """Backward of the 3-NN propagation (etch_weighted_segment_sum_rows over a stable sort of the index list) against fp64 autograd through
    the oracle's feat_propagation (pointnet2_utils.py:45-74); coincident points (zero distance) included; bitwise reproducible."""
    from etch_amd import autograd as A
    from etch_amd import ops
    from oracle import stage1 as S1
    rng = np.random.default_rng(4)
    B, N, S, C = 2, 300, 75, 32
    xyz1 = np.stack([scan(60 + b, N) for b in range(B)])
    xyz2 = xyz1[:, :S].copy()
    feats = rng.standard_normal((B, S, 60, C))
    G = rng.standard_normal((B, N, 60, C))
    p2 = torch.from_numpy(feats).permute(0, 3, 2, 1).reshape(B, C * 60, S).requires_grad_()       # (B, D = c*60+a, S)
    out = S1.feat_propagation(torch.from_numpy(xyz1).double().permute(0, 2, 1), torch.from_numpy(xyz2).double().permute(0, 2, 1), p2)
    (out.view(B, N, C, 60) * torch.from_numpy(G).permute(0, 1, 3, 2)).sum().backward()
    ref = p2.grad.view(B, C, 60, S).permute(0, 3, 2, 1).numpy()
    f = torch.from_numpy(feats.astype(np.float32)).cuda().requires_grad_()
    idx3, w3 = ops.prop3nn(torch.from_numpy(xyz1).cuda(), torch.from_numpy(np.ascontiguousarray(xyz2.transpose(0, 2, 1))).cuda())
    y = A.prop_interp(f, idx3, w3)
    # (the interpolation weights come from fp32 expansion-formula distances, as in the reference's fp32 run: ~2e-5 from the fp64 weights)
    assert rel_err(y.detach().cpu().numpy(), out.detach().view(B, N, C, 60).permute(0, 1, 3, 2).numpy()) < 1e-4
    Gd = torch.from_numpy(G.astype(np.float32)).cuda()
    (y * Gd).sum().backward()
    assert rel_err(f.grad.cpu().numpy(), ref) < 1e-4
    g1 = f.grad.clone()
    f.grad = None
    (A.prop_interp(f, idx3, w3) * Gd).sum().backward()
    assert torch.equal(g1, f.grad)


def test_direction_loss_gradients_of_every_encoder_and_head_parameter(tmp_path):
    """VERDICT r02 task 7: gradients with respect to EVERY parameter of the EPN encoder and the direction head through the model's own
    forward in grad mode (etch_amd.autograd: hand-written backward kernels for the inter / intra SO(3) convs, InstanceNorm + LeakyReLU, the
    3-NN propagation, both attention layers, the linear layers and so3_mean) against autograd through the oracle's restatement in fp64;
    N = 256, B = 2.  Two losses:
      (a) a random linear functional of the anchor weights (the output of so3_reg, models_pointcloud.py:117);
      (b) train.py's cosine direction loss (train.py:80-85).  With seeded random weights the anchor weights are nearly constant over the 60
          anchors and sum_a R_a = 0, so the polar projection sees only their tiny variation: the ORACLE's own fp32 autograd
          lands 1e-2 - 1e-1 from its fp64 run on this loss; it is taken over the points whose projection has a spectral gap (same mask on
          both sides).
    Bar, per parameter tensor, in relative L2 norm: 5e-4, or twice the deviation of the oracle's OWN fp32 autograd from its fp64 run (the
    entitled-error rule of the forward tests).  Why not 1e-4: the composed network is only piecewise differentiable -- of ~10^6 LeakyReLU /
    ReLU inputs per pass about one lies within fp32 rounding of its kink and takes the other slope in a different summation order, which
    moves a weight gradient by ~1e-4 of its norm (measured: torch-CPU fp32 1 - 2e-4, this path 2.6 - 3.8e-4 on loss (a)).  Every backward
    kernel on its own is held to 1e-4 against fp64 in the tests above (convs, InstanceNorm + LeakyReLU, attention, propagation, so3_mean).
    Parameters whose gradient is identically zero (biases in front of an InstanceNorm, the first skip conv on constant occupancy features)
    must come out as numerical zeros."""
    import types

    import torch.nn.functional as F

    from etch_amd import constants as K
    from etch_amd.models.models_pointcloud import GT_network_equiv
    from etch_amd.utils.weights import load_seeded, seeded_state_dict
    from oracle import stage1 as S1
    B, N = 2, 256
    TOL = 5e-4
    args = types.SimpleNamespace(output_folder=str(tmp_path), EPN_input_radius=0.4, EPN_layer_num=2, device=torch.device("cuda"),
                                 markerset=K.default_markerset())
    model = load_seeded(GT_network_equiv(option=args), 1).cuda().eval()
    model.differentiable = True              # eval() mode takes the inference path unless asked (train() mode switches by itself)
    pts = np.stack([scan(80 + b, N) for b in range(B)])
    rng = np.random.default_rng(8)
    vec = rng.standard_normal((B, N, 3))
    vec /= np.linalg.norm(vec, axis=-1, keepdims=True)
    Gaw = rng.standard_normal((B * N, 60))
    names = [k for k, _ in model.named_parameters() if k.startswith(("encoder.", "direction_encoder.", "direction_predictor.", "so3_reg."))]
    assert len(names) == 40
    table = S1.build_layer_table()

    def oracle_grads(dtype, which, mask=None):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            sd = {k: (v.cpu().to(dtype) if v.is_floating_point() else v.cpu()) for k, v in seeded_state_dict(model, 1).items()}
            for k in names:
                sd[k] = sd[k].clone().requires_grad_()
            x = torch.from_numpy(pts).to(dtype)
            xyz, feats = S1.encoder_forward(sd, x, table)
            S_ = xyz.shape[-1]
            pef = S1.feat_propagation(x.permute(0, 2, 1), xyz.to(dtype), feats.permute(0, 1, 3, 2).reshape(B, -1, S_)).reshape(B, N, -1, 60)
            aw = S1.direction_anchor_weights(sd, pef)
            if which == "aw":
                loss = (aw * torch.from_numpy(Gaw).to(dtype)).sum()
            else:
                R, Ce, sv = S1.so3_mean(sd["encoder.backbone.1.blocks.1.intra_conv.conv.anchors"], aw)
                d = R[:, :, 2].reshape(B, N, 3)
                if mask is None:
                    sig = torch.stack([sv[:, 0], sv[:, 1], torch.det(Ce.detach()).sign() * sv[:, 2]], 1).detach()
                    gap = torch.stack([sig[:, 0] + sig[:, 1], sig[:, 0] + sig[:, 2], sig[:, 1] + sig[:, 2]], 1).min(1).values
                    mask = (gap > 0.25 * sv[:, 0].detach()).reshape(B, N)
                m = mask.to(dtype)
                loss = ((1 - F.cosine_similarity(torch.from_numpy(vec).to(dtype), d, dim=-1)) * m).sum() / m.sum()
            loss.backward()
            return {k: sd[k].grad.detach().double().numpy() for k in names}, mask, float(loss.detach()), aw.detach()
        finally:
            torch.set_default_dtype(old)

    def gpu_grads(which, mask=None):
        model.zero_grad(set_to_none=True)
        res, sel = model(torch.from_numpy(pts).cuda(), ["direction"], "standard_vector")
        assert res["direction"].requires_grad and sel.shape == (B, N, 3)
        if which == "aw":
            loss = (model.last_anc_w.reshape(B * N, 60) * torch.from_numpy(Gaw.astype(np.float32)).cuda()).sum()
        else:
            m = mask.float().cuda()
            loss = ((1 - F.cosine_similarity(torch.from_numpy(vec.astype(np.float32)).cuda(), res["direction"], dim=-1)) * m).sum() / m.sum()
        loss.backward()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if k in names}, float(loss.detach())

    def compare(gg, g64, g32, tag, tol=1e-4, slack=2.0):
        top = max(np.abs(v).max() for v in g64.values())
        rows = []
        for k in names:
            assert gg[k] is not None, k
            mine = gg[k].cpu().double().numpy().reshape(g64[k].shape)
            scale = np.abs(g64[k]).max()
            if scale < 1e-6 * top:                                       # identically zero gradient: a numerical zero, not a ratio
                assert np.abs(mine).max() <= max(1e-4 * top, 4.0 * np.abs(g32[k]).max()), (tag, k, np.abs(mine).max(), np.abs(g32[k]).max())
                continue
            l2 = lambda a: float(np.linalg.norm((a - g64[k]).ravel()) / np.linalg.norm(g64[k].ravel()))
            rows.append((l2(mine), l2(g32[k]), float(np.abs(mine - g64[k]).max() / scale), float(np.abs(g32[k] - g64[k]).max() / scale), k))
        rows.sort(reverse=True)
        print(f"{tag}: {len(rows)} tensors with a gradient; deviation from the fp64 oracle, relative L2 (gpu / oracle fp32) and max-norm (gpu / oracle fp32):")
        for r in rows[:8]:
            print("   %-64s L2 %.1e / %.1e   max %.1e / %.1e" % (r[4], r[0], r[1], r[2], r[3]))
        for e_gpu, e_ref, m_gpu, m_ref, k in rows:
            assert e_gpu <= max(tol, slack * e_ref), (tag, k, e_gpu, e_ref)
        return len(rows)

    # (a) linear functional of the anchor weights: strict
    g64, _, l64, aw64 = oracle_grads(torch.float64, "aw")
    g32, _, _, _ = oracle_grads(torch.float32, "aw")
    gg, lg = gpu_grads("aw")
    assert rel_err(model.last_anc_w.detach().cpu().numpy().reshape(-1, 60), aw64.numpy()) < 1e-4
    assert abs(lg - l64) < 1e-4 * max(1.0, np.abs(Gaw).sum() * float(aw64.abs().max()))
    assert compare(gg, g64, g32, "anchor-weight functional", tol=TOL) >= 25
    g2, _ = gpu_grads("aw")
    for k in names:
        assert torch.equal(g2[k], gg[k]), k                              # bitwise reproducible
    # (b) the cosine direction loss over the well-gapped points: as close to fp64 as the oracle's own fp32 autograd (x2)
    c64, mask, lc64, _ = oracle_grads(torch.float64, "cos")
    assert 0.15 < float(mask.float().mean()) < 1.0
    c32, _, lc32, _ = oracle_grads(torch.float32, "cos", mask)
    cg, lcg = gpu_grads("cos", mask)
    print("masked cosine loss: fp64 oracle %.6f, fp32 oracle %.6f, gpu %.6f" % (lc64, lc32, lcg))
    # the loss itself inherits the conditioning of the projection (a 1e-4 deviation of the anchor weights moves a direction by up to
    # 2 |dCe| / gap, tests/_parity.py:direction_within_conditioning): a loose sanity bound, the gradients are compared below
    assert abs(lcg - lc64) <= max(2e-2, 2.0 * abs(lc32 - lc64))
    # end-to-end sanity only: on this loss fp32 is chaotic (the oracle's own fp32 autograd: 4e-2 relative L2 on the first conv), what pins the
    # chain is (a) above (d anchor-weights / d parameters) together with the so3_mean backward test (d loss / d anchor-weights)
    compare(cg, c64, c32, "cosine direction loss", tol=TOL, slack=8.0)
    # untouched heads get no gradient; in no_grad mode the same call takes the fused inference path
    assert all(p.grad is None for k, p in model.named_parameters() if k.startswith(("confidence_encoder.", "magnitude_encoder.")))
    with torch.no_grad():
        res0, _ = model(torch.from_numpy(pts).cuda(), ["direction"], "standard_vector")
    assert not res0["direction"].requires_grad
    model.differentiable = None              # default: eval() mode -> inference path even with gradients enabled (a forgotten no_grad)
    res1, _ = model(torch.from_numpy(pts).cuda(), ["direction"], "standard_vector")
    assert not res1["direction"].requires_grad and torch.equal(res1["direction"], res0["direction"])
    assert rel_err(model.last_anc_w.cpu().numpy().reshape(-1, 60), aw64.numpy()) < 1e-4
