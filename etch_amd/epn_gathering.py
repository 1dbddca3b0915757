"""Drop-in for the reference's pybind module `epn_gathering` (external/vgtk/vgtk/cuda/gathering_cuda.cpp:60-64)."""
from .ops import gather_points_forward  # noqa: F401


def gather_points_backward(*a, **k):
    raise NotImplementedError("training-only (SURVEY 8f-3)")
