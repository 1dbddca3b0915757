"""Drop-in for the reference's pybind module `pointops_cuda` (external/pointops/src/pointops_api.cpp:12-23):
same names, same caller-allocated-output convention, for the two functions on the hot path."""
import ctypes

import torch

from . import _lib
from .ops import _ptr, _stream


def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    """Fills idx (m,nsample) int32 and dist2 (m,nsample) with SQUARED distances (knnquery_cuda.cpp:9-19)."""
    # segments are found on the device like knnquery_cuda_kernel.cu:52-62,74-80: no host read of the offsets
    # (the number of segments is the offsets tensor's element count -- known here without a sync: the kernel's segment scan is clamped to it)
    _lib.check(_lib.lib().etch_knnquery_dev_bounded(int(m), int(nsample), int(new_offset.numel()), _ptr(xyz), _ptr(new_xyz), _ptr(offset), _ptr(new_offset),
                                                    _ptr(idx), _ptr(dist2), 0, _stream()), "etch_knnquery_dev_bounded")


def furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):
    """Fills idx (sampling_cuda.cpp:9-18).  `tmp` is accepted for signature compatibility and unused."""
    _lib.check(_lib.lib().etch_furthestsampling(int(b), int(n_max), _ptr(xyz), _ptr(offset), _ptr(new_offset), _ptr(idx), _stream()),
               "etch_furthestsampling")
