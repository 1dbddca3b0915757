"""MI355X counterpart of /root/reference/external/vgtk/vgtk/so3conv/functional.py: the same function names, argument meaning and
tensor shapes, every op a HIP kernel behind the C ABI (csrc/functional_ops.hip).  Reached as `etch_amd.vgtk_so3conv.functional`,
the way the reference spells `vgtk.so3conv.functional`.

These are the UN-FUSED forms: they materialise the tensors the reference materialises (the [b,p,60,24,nn] kernel weights are 921 MB
per 5 000-point scan at the first conv).  The model itself never calls them -- `InterSO3Conv` / `IntraSO3Conv` run the fused
kernels -- they serve callers of the operator API and cross-check the fused kernels in tests/test_gpu_functional.py."""
import ctypes
import math

import torch

from . import _lib
from . import constants as K
from . import ops
from .ops import _c_float, _need, _ptr, _stream


def batched_index_select(input, dim, index):
    """functional.py:51-59 for the one pattern the path uses (feats [b,c,q,a], dim = 2, index [b,m]) -> [b,c,m,a]."""
    assert input.dim() == 4 and dim == 2 and index.dim() == 2
    b, c, q, a = input.shape
    rows = input.permute(0, 2, 1, 3).reshape(b, q, c * a).contiguous()
    out = index_points(rows, index.long())
    return out.view(b, -1, c, a).permute(0, 2, 1, 3).contiguous()


def index_points(points, idx):
    """src/models/pointnet2_utils.py:26-43: points [B,N,C], idx [B,S...] (long) -> [B,S...,C]."""
    points = points.contiguous()
    _need(points, torch.float32, "points")
    idx = idx.long().contiguous()
    B, N, C = points.shape
    S = idx.numel() // B
    out = torch.empty((B, S, C), dtype=torch.float32, device=points.device)
    _lib.check(_lib.lib().etch_index_points(B, N, ctypes.c_long(S), C, _ptr(points), _ptr(idx), _ptr(out), _stream()), "etch_index_points")
    return out.view(*idx.shape, C)


def square_distance(src, dst):
    """src/models/pointnet2_utils.py:4-23: [B,N,C] x [B,M,C] -> [B,N,M]."""
    src, dst = src.contiguous(), dst.contiguous()
    _need(src, torch.float32, "src"), _need(dst, torch.float32, "dst")
    B, N, C = src.shape
    M = dst.shape[1]
    out = torch.empty((B, N, M), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().etch_square_distance(B, N, M, C, _ptr(src), _ptr(dst), _ptr(out), _stream()), "etch_square_distance")
    return out


def add_shadow_point(x):
    """functional.py:93-97: [b,c,n] -> [b,c,n+1] with a far-away point (1e4)."""
    b, c, _ = x.shape
    return torch.cat((x, torch.full((b, c, 1), 1e4, dtype=x.dtype, device=x.device)), dim=2).contiguous()


def add_shadow_feature(x):
    """functional.py:101-105: [b,c,n,a] -> [b,c,n+1,a] with a zero row."""
    b, c, _, a = x.shape
    return torch.cat((x, torch.zeros((b, c, 1, a), dtype=x.dtype, device=x.device)), dim=2).contiguous()


def get_occupancy_features(pc, n_anchor, use_center=False):
    """functional.py:70-89 (points without normals): [nb,np,3] -> ones [nb,1,np,na]."""
    nb, npnt, nd = pc.shape
    assert nd == 3, "normals are not used on the ETCH path"
    f = torch.ones((nb, 1, npnt, n_anchor), dtype=torch.float32, device=pc.device)
    if use_center:
        f[:, :, 0, :] = 0.0
    return f


def get_sphereical_kernel_points_from_ply(radius, kernel_size):
    """functional.py:146-157 (kernel_size = 1 -> the 24 points of kpsphere24.ply scaled so that the farthest one sits at `radius`;
    InterSO3Conv passes KERNEL_CONDENSE_RATIO * conv radius, modules.py:99)."""
    return K.get_kernel_points(radius / K.KERNEL_CONDENSE_RATIO, kernel_size)


def ball_query(query_points, support_points, radius, n_sample, support_feats=None):
    """functional.py:159-166 -> idx, grouped xyz [b,3,m,ns] (, grouped feats)."""
    from . import vgtk_so3conv as V
    idx = V.ball_query_index(query_points, support_points, radius, n_sample)
    sp = add_shadow_point(support_points)
    if support_feats is None:
        return idx, V.group_nd(sp, idx)
    return idx, V.group_nd(sp, idx), V.group_nd(support_feats, idx)


def inter_spconv_grouping_ball(xyz, stride, radius, n_neighbor, lazy_sample=True):
    """functional.py:176-185 -> grouped_xyz [b,3,p2,nn] (relative), ball_idx [b,p2,nn], sample idx [b,p2], new_xyz [b,3,p2]."""
    from . import vgtk_so3conv as V
    n_sample = math.ceil(xyz.shape[2] / stride)
    idx, sample_xyz = V.furthest_sample(xyz, n_sample, lazy_sample)
    ball_idx, grouped_xyz = ball_query(sample_xyz, xyz, radius, n_neighbor)
    return grouped_xyz - sample_xyz.unsqueeze(3), ball_idx, idx, sample_xyz


def inter_so3conv_grouping_anchor(grouped_xyz, anchors, kernels, sigma, interpolate="linear"):
    """functional.py:286-324: grouped_xyz [b,3,p2,nn], anchors [na,3,3], kernels [ks,3] -> inter_w [b,p2,na,ks,nn]."""
    if interpolate != "linear":
        raise NotImplementedError("kernel function %s is not implemented!" % interpolate)
    grouped_xyz = grouped_xyz.contiguous()
    _need(grouped_xyz, torch.float32, "grouped_xyz")
    b, _, p, nn = grouped_xyz.shape
    na, ks = anchors.shape[0], kernels.shape[0]
    # rotated kernel points exactly as functional.py:296 (host matmul, then uploaded), laid out [na, ks, 3]
    rk = torch.matmul(anchors.detach().cpu(), kernels.detach().cpu().transpose(0, 1)).permute(0, 2, 1).contiguous().to(grouped_xyz.device)
    w = torch.empty((b, p, na, ks, nn), dtype=torch.float32, device=grouped_xyz.device)
    _lib.check(_lib.lib().etch_inter_kernel_weights(b, p, nn, na, ks, _ptr(grouped_xyz), _ptr(rk), _c_float(sigma), _ptr(w), _stream()),
               "etch_inter_kernel_weights")
    return w


def inter_so3conv_feat_grouping(inter_idx, inter_w, feats):
    """functional.py:61-67: inter_idx [b,p,nn], inter_w [b,p,na,ks,nn], feats [b,c,q,na] (shadow-padded) -> [b,c,ks,p,na]."""
    feats, inter_w = feats.contiguous(), inter_w.contiguous()
    idx = inter_idx.int().contiguous()
    _need(feats, torch.float32, "feats"), _need(inter_w, torch.float32, "inter_w")
    b, p, nn = idx.shape
    _, c, q, na = feats.shape
    ks = inter_w.shape[3]
    out = torch.empty((b, c, ks, p, na), dtype=torch.float32, device=feats.device)
    _lib.check(_lib.lib().etch_inter_feat_grouping(b, c, q, p, nn, na, ks, _ptr(idx), _ptr(inter_w), _ptr(feats), _ptr(out), _stream()),
               "etch_inter_feat_grouping")
    return out


def inter_so3conv_grouping(xyz, feats, stride, n_neighbor, anchors, kernels, radius, sigma, inter_idx=None, inter_w=None, lazy_sample=True,
                           radius_expansion=1.0, pooling=None):
    """functional.py:224-284 (pooling=None, the ETCH configuration) -> inter_idx, inter_w, new_xyz, new_feats [b,c,ks,p2,na], sample_idx."""
    if pooling is not None:
        raise NotImplementedError("ETCH builds every conv with pooling=None (models_pointcloud.py:46-48)")
    if inter_idx is None:
        grouped_xyz, inter_idx, sample_idx, new_xyz = inter_spconv_grouping_ball(xyz, stride, radius * radius_expansion, n_neighbor, lazy_sample)
        inter_w = inter_so3conv_grouping_anchor(grouped_xyz, anchors, kernels, sigma)
    else:
        sample_idx, new_xyz = None, xyz
    new_feats = inter_so3conv_feat_grouping(inter_idx, inter_w, add_shadow_feature(feats))
    return inter_idx, inter_w, new_xyz, new_feats, sample_idx


def intra_so3conv_grouping(intra_idx, feature):
    """functional.py:331-378: intra_idx [na,pnn] (long), feature [nb,c,np,na] -> [nb,c,pnn,np,na]."""
    feature = feature.contiguous()
    _need(feature, torch.float32, "feature")
    ii = intra_idx.long().contiguous()
    nb, c, nq, na = feature.shape
    pnn = ii.shape[1]
    out = torch.empty((nb, c, pnn, nq, na), dtype=torch.float32, device=feature.device)
    _lib.check(_lib.lib().etch_intra_grouping(nb, c, nq, na, pnn, _ptr(ii), _ptr(feature), _ptr(out), _stream()), "etch_intra_grouping")
    return out


def get_anchors(k=60):
    return K.get_anchors(k)


def get_intra_idx():
    return K.get_intra_idx()
