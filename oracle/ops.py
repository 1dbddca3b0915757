"""ctypes front-end of oracle/discrete_ops.c (CPU restatement of the reference's CUDA index kernels).

TEST INFRASTRUCTURE ONLY.  Function names mirror the reference pybind modules:
  epn_grouping.{ball_query, furthest_point_sampling}   external/vgtk/vgtk/cuda/grouping_cuda.cpp:71-86,158-173
  epn_gathering.gather_points_forward                  external/vgtk/vgtk/cuda/gathering_cuda.cpp:29-46
  pointops_cuda.{knnquery_cuda, furthestsampling_cuda} external/pointops/src/pointops_api.cpp:12-14
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_ops.so")
_lib = None
_variants = {}


def build(force=False):
    src_t = os.path.getmtime(os.path.join(_HERE, "discrete_ops.c"))
    sos = [_SO] + [os.path.join(_HERE, "_build", f"liboracle_ops_fma{k}.so") for k in (1, 2)]
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < src_t for p in sos):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_opt_n_threads.restype = ctypes.c_int
    return _lib


class variant:
    """Context manager: route every op of this module through liboracle_ops_fma<k>.so (k = 1, 2: the two plausible nvcc
    contractions of the squared distance, discrete_ops.c ORC_FMA).  Tests only."""

    def __init__(self, k):
        self.k = int(k)

    def __enter__(self):
        global _lib
        lib()
        self.prev = _lib
        if self.k:
            if self.k not in _variants:
                v = ctypes.CDLL(os.path.join(_HERE, "_build", f"liboracle_ops_fma{self.k}.so"))
                v.orc_opt_n_threads.restype = ctypes.c_int
                _variants[self.k] = v
            _lib = _variants[self.k]
        return self

    def __exit__(self, *a):
        global _lib
        _lib = self.prev


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def opt_n_threads(n):
    return int(lib().orc_opt_n_threads(ctypes.c_int(int(n))))


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (b,3,m), xyz (b,3,n) -> idx (b,m,nsample) int32."""
    new_xyz, xyz = _f32(new_xyz), _f32(xyz)
    b, _, m = new_xyz.shape
    n = xyz.shape[2]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().orc_ball_query(b, n, m, ctypes.c_float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx))
    return idx


def furthest_point_sampling(xyz, m, keyed=False):
    """xyz (b,3,n) -> idx (b,m) int32 (vgtk flavour: origin skip, start index 0)."""
    xyz = _f32(xyz)
    b, _, n = xyz.shape
    idx = np.zeros((b, m), np.int32)
    fn = lib().orc_fps_vgtk_keyed if keyed else lib().orc_fps_vgtk
    fn(b, n, m, _p(xyz), _p(idx))
    return idx


def gather_points_forward(points, idx):
    """points (b,c,n), idx (b,m) -> (b,c,m)."""
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), np.float32)
    lib().orc_gather_points(b, c, n, m, _p(points), _p(idx), _p(out))
    return out


def gather_points_backward(grad_out, idx, npoint):
    """grad_out (b,c,m), idx (b,m) -> grad_points (b,c,n)."""
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m = grad_out.shape
    out = np.empty((b, c, int(npoint)), np.float32)
    lib().orc_gather_points_backward(b, c, int(npoint), m, _p(grad_out), _p(idx), _p(out))
    return out


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """xyz (n,3), new_xyz (m,3), offsets (b) -> idx (m,nsample) int32, dist2 (m,nsample) (SQUARED)."""
    assert nsample <= 100
    xyz, new_xyz, offset, new_offset = _f32(xyz), _f32(new_xyz), _i32(offset), _i32(new_offset)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), np.int32)
    d2 = np.zeros((m, nsample), np.float32)
    lib().orc_knnquery(len(offset), m, nsample, _p(xyz), _p(new_xyz), _p(offset), _p(new_offset), _p(idx), _p(d2))
    return idx, d2


def furthestsampling(xyz, offset, new_offset):
    """xyz (n,3), offset (b), new_offset (b) -> idx (new_offset[-1]) int32 (pointops flavour)."""
    xyz, offset, new_offset = _f32(xyz), _i32(offset), _i32(new_offset)
    b = len(offset)
    seg = np.diff(np.concatenate([[0], offset]))
    n_max = int(seg.max())
    idx = np.zeros((int(new_offset[-1]),), np.int32)
    lib().orc_fps_pointops(b, n_max, xyz.shape[0], _p(xyz), _p(offset), _p(new_offset), _p(idx))
    return idx
