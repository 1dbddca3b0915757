"""Architecture constants of the EPN encoder (60 icosahedral rotation anchors, 60x12 intra-neighbour
index, 24 KPConv kernel points) and the 86-marker SMPL superset.

The numbers are the outputs of the reference's generators, captured once by
oracle/ref_harness/gen_golden.py and shipped as data (etch_amd/data/epn_constants.npz):
  anchors / intra_idx : vgtk/functional/rotation.py:237-345 (icosahedron_so3_trimesh) via functional.py:387-405
  kp24_raw            : vgtk/data/anchors/kpsphere24.ply read by functional.py:146-157
  marker_vids/names   : datafolder/useful_data_4d-dress/superset_smpl.json
Reference checkpoints carry anchors / kernels / intra_idx as buffers and override these on load.
"""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "epn_constants.npz")
_cache = None
KERNEL_CONDENSE_RATIO = 0.7  # modules.py:13


def _c():
    global _cache
    if _cache is None:
        _cache = dict(np.load(_PATH, allow_pickle=False))
    return _cache


def get_anchors(k=60):
    a = _c()["anchors"]
    if k == 60:
        return a.copy()
    if k == 1:
        return a[29][None].copy()
    if k == 20:
        return a[::3].copy()
    raise ValueError(k)


def get_intra_idx():
    return _c()["intra_idx"].copy()


def get_kernel_points(radius, kernel_size=1):
    """functional.py:146-157 `get_sphereical_kernel_points_from_ply(0.7*radius, 1)`: same fp32 arithmetic."""
    assert kernel_size == 1, "only kpsphere24 (kernel_size=1) is used on the ETCH path (so3net.py:80)"
    kp = _c()["kp24_raw"]
    r = np.sqrt((kp ** 2).sum(1).max())
    return (kp * (KERNEL_CONDENSE_RATIO * radius) / r).astype(np.float32)


def default_markerset():
    c = _c()
    return {str(n): int(v) for n, v in zip(c["marker_names"], c["marker_vids"])}
