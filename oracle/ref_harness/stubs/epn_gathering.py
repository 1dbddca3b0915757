"""CPU stand-in for the reference's CUDA pybind module `epn_gathering` (oracle-backed).  fp64 inputs (only used by the
conditioning study gen_golden.gen_padding_fp64, where the whole reference model is run in double) are gathered by numpy
so the copy keeps its dtype; the fp32 path is the C restatement."""
import numpy as np
import torch

from oracle import ops as _o


def gather_points_forward(points, idx):
    if points.dtype == torch.float64:
        i = idx.numpy().astype(np.int64)
        return torch.from_numpy(np.take_along_axis(points.numpy(), i[:, None, :], axis=2))
    return torch.from_numpy(_o.gather_points_forward(points.numpy(), idx.numpy()))
