"""CPU stand-in for the reference's CUDA pybind module `epn_grouping`, backed by the oracle's C
restatement (oracle/discrete_ops.c).  Only used to run the reference Python on CPU in this container."""
import torch

from oracle import ops as _o


def ball_query(new_xyz, xyz, radius, nsample):
    return torch.from_numpy(_o.ball_query(new_xyz.numpy(), xyz.numpy(), float(radius), int(nsample)))


def furthest_point_sampling(xyz, m):
    return torch.from_numpy(_o.furthest_point_sampling(xyz.numpy(), int(m)))


def initial_anchor_query(*a, **k):
    raise NotImplementedError("not on the hot path")


def anchor_query(*a, **k):
    raise NotImplementedError("not on the hot path")
