"""CPU stand-in for the reference's CUDA pybind module `epn_gathering` (oracle-backed)."""
import torch

from oracle import ops as _o


def gather_points_forward(points, idx):
    return torch.from_numpy(_o.gather_points_forward(points.numpy(), idx.numpy()))
