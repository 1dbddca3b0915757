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
