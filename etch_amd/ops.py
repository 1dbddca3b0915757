"""Torch-tensor front-end of the C ABI (include/etch_hip.h).

PyTorch is plumbing only: it owns device memory and the current HIP stream.  Every function here
hands raw device pointers to libetch_hip.so; there is no eager/CPU fallback -- a CPU tensor or a
missing library raises.
"""
import ctypes

import torch

from . import _lib

_c_int, _c_float, _vp = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return _vp(t.data_ptr())


def _need(t, dtype, name):
    if not t.is_cuda:
        raise _lib.EtchHipError(f"{name} must be a CUDA(HIP) tensor: etch_amd has no CPU path")
    if t.dtype != dtype:
        raise _lib.EtchHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.EtchHipError(f"{name} must be contiguous")  # grouping_cuda.cpp:66-68 CHECK_INPUT
    return t


# ------------------------------------------------------------------ epn_grouping / epn_gathering
def ball_query(new_xyz, xyz, radius, nsample):
    """epn_grouping.ball_query (grouping_cuda.cpp:71-86): (b,3,m),(b,3,n) -> idx (b,m,nsample) int32."""
    _need(new_xyz, torch.float32, "new_xyz"), _need(xyz, torch.float32, "xyz")
    b, _, m = new_xyz.shape
    n = xyz.shape[2]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz.device)
    _lib.check(_lib.lib().etch_ball_query(b, n, m, _c_float(radius), int(nsample), _ptr(new_xyz), _ptr(xyz), _ptr(idx), _stream()),
               "etch_ball_query")
    return idx


def furthest_point_sampling(xyz, m):
    """epn_grouping.furthest_point_sampling (grouping_cuda.cpp:158-173): (b,3,n) -> (b,m) int32."""
    _need(xyz, torch.float32, "xyz")
    b, _, n = xyz.shape
    idx = torch.zeros((b, m), dtype=torch.int32, device=xyz.device)
    _lib.check(_lib.lib().etch_furthest_point_sampling(b, n, int(m), _ptr(xyz), _ptr(idx), _stream()), "etch_furthest_point_sampling")
    return idx


def gather_points_forward(points, idx):
    """epn_gathering.gather_points_forward (gathering_cuda.cpp:29-46): (b,c,n),(b,m) -> (b,c,m)."""
    _need(points, torch.float32, "points"), _need(idx, torch.int32, "idx")
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    _lib.check(_lib.lib().etch_gather_points(b, c, n, m, _ptr(points), _ptr(idx), _ptr(out), _stream()), "etch_gather_points")
    return out


# ------------------------------------------------------------------ pointops_cuda
def _seg_max(offset_host):
    prev, mx = 0, 0
    for v in offset_host:
        mx = max(mx, int(v) - prev)
        prev = int(v)
    return mx


def knnquery(nsample, xyz, new_xyz, offset, new_offset, new_offset_host=None, sqrt=True):
    """pointops.knnquery (pointops.py:32-45): -> idx (m,nsample) int32, dist (m,nsample) = sqrt(d2).

    `new_offset_host` (list of ints) avoids a device->host sync for the grid size; if omitted it is
    read back from `new_offset` (one sync, like the reference's own .item() calls)."""
    if new_xyz is None:
        new_xyz = xyz
    _need(xyz, torch.float32, "xyz"), _need(new_xyz, torch.float32, "new_xyz")
    _need(offset, torch.int32, "offset"), _need(new_offset, torch.int32, "new_offset")
    if new_offset_host is None:
        new_offset_host = new_offset.tolist()
    m = new_xyz.shape[0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    dist = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    _lib.check(_lib.lib().etch_knnquery(len(new_offset_host), _seg_max(new_offset_host), int(nsample), _ptr(xyz), _ptr(new_xyz),
                                        _ptr(offset), _ptr(new_offset), _ptr(idx), _ptr(dist), 1 if sqrt else 0, _stream()),
               "etch_knnquery")
    return idx, dist


def furthestsampling(xyz, offset, new_offset, offset_host=None, new_offset_host=None):
    """pointops.furthestsampling (pointops.py:10-28): packed (n,3) + offsets -> idx (m) int32."""
    _need(xyz, torch.float32, "xyz"), _need(offset, torch.int32, "offset"), _need(new_offset, torch.int32, "new_offset")
    if offset_host is None:
        offset_host = offset.tolist()
    if new_offset_host is None:
        new_offset_host = new_offset.tolist()
    idx = torch.zeros((int(new_offset_host[-1]),), dtype=torch.int32, device=xyz.device)
    _lib.check(_lib.lib().etch_furthestsampling(len(offset_host), _seg_max(offset_host), _ptr(xyz), _ptr(offset), _ptr(new_offset),
                                                _ptr(idx), _stream()), "etch_furthestsampling")
    return idx
