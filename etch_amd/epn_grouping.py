"""Drop-in for the reference's pybind module `epn_grouping` (external/vgtk/vgtk/cuda/grouping_cuda.cpp:176-181),
backed by libetch_hip.so.  Only the functions on the hot path exist."""
from .ops import ball_query, furthest_point_sampling  # noqa: F401


def initial_anchor_query(*a, **k):
    raise NotImplementedError("epn_grouping.initial_anchor_query is not on the ETCH inference path (SURVEY 2.2)")


def anchor_query(*a, **k):
    raise NotImplementedError("epn_grouping.anchor_query is not on the ETCH inference path (SURVEY 2.2)")
