"""CPU stand-in for the reference's CUDA pybind module `pointops_cuda` (oracle-backed).
Caller-allocated outputs, as in external/pointops/src/pointops_api.cpp:12-23."""
import torch

from oracle import ops as _o


def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    i, d = _o.knnquery(int(nsample), xyz.numpy(), new_xyz.numpy(), offset.numpy(), new_offset.numpy())
    idx.copy_(torch.from_numpy(i))
    dist2.copy_(torch.from_numpy(d))


def furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):
    idx.copy_(torch.from_numpy(_o.furthestsampling(xyz.numpy(), offset.numpy(), new_offset.numpy())))
