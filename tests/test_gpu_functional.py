"""GPU tests of the reference's operator names that sit beside the fused hot path: vgtk.so3conv.functional.*,
pointnet2_utils.square_distance / index_points, models.so3conv.so3_mean.  Each un-fused form is checked against the oracle's
restatement (itself pinned by the reference goldens) and, chained, against the fused kernel the model actually runs."""
import numpy as np
import pytest
import torch

from _parity import rel_err

pytestmark = pytest.mark.gpu


def scan(seed, n):
    return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)


@pytest.mark.parametrize("stride,lazy,nn,cin,cout", [(2, False, 16, 16, 32), (1, True, 24, 32, 32)])
def test_inter_grouping_unfused_forms_vs_oracle_and_fused_kernel(stride, lazy, nn, cin, cout):
    import etch_amd.vgtk_so3conv as V
    from etch_amd.utils.weights import load_seeded
    from oracle import stage1 as S1
    L = V.functional
    b, n, radius, sigma = 2, 300, 0.2, 0.02
    rng = np.random.default_rng(0)
    xyz = torch.from_numpy(np.stack([scan(20 + i, n).T.copy() for i in range(b)]))
    feats = torch.from_numpy(rng.standard_normal((b, cin, n, 60)).astype(np.float32))
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, stride, radius, sigma, nn, lazy_sample=lazy), 4).cuda().eval()
    anchors, kernels = conv.anchors, conv.kernels
    # oracle (reference semantics, functional.py:176-185, 286-324, 61-67)
    g_ref, ball_ref, sidx_ref, new_ref = S1.inter_grouping(xyz, stride, radius, nn, lazy)
    w_ref = S1.inter_weights(g_ref, anchors.cpu(), kernels.cpu(), sigma)
    nf_ref = S1.inter_feat_grouping(ball_ref, w_ref, torch.cat((feats, torch.zeros(b, cin, 1, 60)), 2).contiguous())
    # un-fused HIP forms under the reference's names
    g, ball, sidx, new_xyz = L.inter_spconv_grouping_ball(xyz.cuda(), stride, radius, nn, lazy)
    assert torch.equal(ball.cpu(), ball_ref) and torch.equal(sidx.cpu(), sidx_ref) and torch.equal(new_xyz.cpu(), new_ref)
    assert torch.equal(g.cpu(), g_ref)
    w = L.inter_so3conv_grouping_anchor(g, anchors, kernels, sigma)
    assert w.shape == (b, new_xyz.shape[2], 60, 24, nn)
    assert float((w.cpu() - w_ref).abs().max()) < 2e-6
    assert float(((w.cpu() > 0) != (w_ref > 0)).float().mean()) < 1e-5           # the exact zeros sit in the same places
    idx2, w2, new2, nf, sidx2 = L.inter_so3conv_grouping(xyz.cuda(), feats.cuda(), stride, nn, anchors, kernels, radius, sigma, lazy_sample=lazy)
    assert nf.shape == (b, cin, 24, new_xyz.shape[2], 60) and torch.equal(idx2, ball)
    assert rel_err(nf.cpu().numpy(), nf_ref.numpy()) < 1e-5
    # chained with BasicSO3Conv (modules.py:33-39) it is the fused kernel's result
    y_unfused = conv.basic_conv(nf)
    _, _, _, cloud = conv(V.SphericalPointCloud(xyz.cuda(), feats.cuda(), anchors))
    assert rel_err(cloud.feats.cpu().numpy(), y_unfused.cpu().numpy()) < 2e-5
    y_ref = S1.basic_so3conv(conv.basic_conv.W.detach().cpu(), conv.basic_conv.bias.detach().cpu(), nf_ref)
    assert rel_err(cloud.feats.cpu().numpy(), y_ref.numpy()) < 1e-4


def test_intra_grouping_vs_reference_form_and_fused_kernel():
    import etch_amd.vgtk_so3conv as V
    from etch_amd.utils.weights import load_seeded
    L = V.functional
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((2, 32, 70, 60)).astype(np.float32))
    ii = torch.from_numpy(L.get_intra_idx()).long()
    want = x.index_select(3, ii.view(-1)).view(2, 32, 70, 60, 12).permute(0, 1, 4, 2, 3).contiguous()      # functional.py:343-344
    got = L.intra_so3conv_grouping(ii.cuda(), x.cuda())
    assert torch.equal(got.cpu(), want)
    conv = load_seeded(V.IntraSO3Conv(32, 32), 5).cuda().eval()
    y_unfused = conv.basic_conv(got)
    y = conv(V.SphericalPointCloud(None, x.cuda(), conv.anchors)).feats
    assert rel_err(y.cpu().numpy(), y_unfused.cpu().numpy()) < 2e-5


def test_square_distance_and_index_points():
    from etch_amd.models.pointnet2_utils import index_points, square_distance
    from oracle import stage1 as S1
    rng = np.random.default_rng(2)
    src = torch.from_numpy(rng.standard_normal((2, 130, 3)).astype(np.float32))
    dst = torch.cat([src[:, :40], torch.from_numpy(rng.standard_normal((2, 37, 3)).astype(np.float32))], 1)   # coincident points: d ~ 0
    d = square_distance(src.cuda(), dst.cuda())
    ref = S1.square_distance(src, dst)
    assert d.shape == (2, 130, 77) and float((d.cpu() - ref).abs().max()) < 1e-6 * float(ref.abs().max())   # CPU matmul sums in another order
    pts = torch.from_numpy(rng.standard_normal((2, 50, 7)).astype(np.float32))
    for shape in ((2, 9), (2, 5, 3)):
        idx = torch.from_numpy(rng.integers(0, 50, shape))
        got = index_points(pts.cuda(), idx.cuda())
        want = torch.stack([pts[b][idx[b]] for b in range(2)])
        assert got.shape == want.shape and torch.equal(got.cpu(), want)


def test_so3_mean_reference_signature(golden):
    from etch_amd.models.so3conv import so3_mean
    from oracle import stage1 as S1
    g, c = golden("module_direction.npz"), golden("constants.npz")
    A = torch.from_numpy(c["anchors"]).cuda()
    w = torch.from_numpy(g["mean_w"]).cuda()
    T = w.shape[0]
    R = so3_mean(A[None].expand(T, -1, -1, -1), w)                       # the model's call pattern: one shared set (models_pointcloud.py:118)
    assert R.shape == (T, 3, 3) and np.abs(R.cpu().numpy() - g["mean_R"]).max() < 1e-5
    R2 = so3_mean(A[None].repeat(T, 1, 1, 1), w)                         # materialised per-row copies: same result
    assert torch.equal(R, R2)
    # per-row rotation sets, no weights: chordal mean of rotations clustered around a known one
    rng = np.random.default_rng(3)
    base = c["anchors"][rng.integers(0, 60, 16)]
    noise = torch.from_numpy(rng.standard_normal((16, 7, 3)).astype(np.float32) * 0.1)
    from oracle import stage2 as S2
    Rs = torch.from_numpy(base)[:, None] @ S2.rodrigues(noise.view(-1, 3)).view(16, 7, 3, 3)
    got = so3_mean(Rs.cuda()).cpu()
    Ce = Rs.double().sum(1)
    U, _, Vh = torch.linalg.svd(Ce)
    D = torch.diag_embed(torch.stack([torch.ones(16), torch.ones(16), torch.det(U @ Vh)], 1).double())
    assert float((got.double() - U @ D @ Vh).abs().max()) < 1e-5
    assert float((got @ got.transpose(1, 2) - torch.eye(3)).abs().max()) < 1e-5
