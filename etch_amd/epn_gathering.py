"""Drop-in for the reference's pybind module `epn_gathering` (external/vgtk/vgtk/cuda/gathering_cuda.cpp:60-64)."""
from .ops import gather_points_backward, gather_points_forward  # noqa: F401
