"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/etch_hip.h declares.  No compute call is made (no GPU here)."""
import os

from etch_amd import _lib
from etch_amd import build as B


def test_library_builds_and_exports_header_symbols():
    lib_path = B.build()
    assert os.path.exists(lib_path)
    names = _lib.declared_symbols()
    assert len(names) >= 5 and "etch_ball_query" in names
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), n


def test_product_has_no_cpu_fallback():
    import pytest
    import torch

    from etch_amd import ops

    with pytest.raises(_lib.EtchHipError):
        ops.ball_query(torch.zeros(1, 3, 4), torch.zeros(1, 3, 8), 0.1, 4)


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "etch_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, f)


def _norm(proto):
    import re
    proto = re.sub(r"/\*.*?\*/", "", proto, flags=re.S)
    return re.sub(r"\s+", " ", proto.replace("*", " * ")).replace(" ;", ";").strip()


REFERENCE_LAUNCHERS = {
    # external/pointops/src/knnquery/knnquery_cuda_kernel.h:14
    "knnquery_cuda_launcher": "void knnquery_cuda_launcher(int m, int nsample, const float *xyz, const float *new_xyz, const int *offset, "
                              "const int *new_offset, int *idx, float *dist2);",
    # external/pointops/src/sampling/sampling_cuda_kernel.h:14
    "furthestsampling_cuda_launcher": "void furthestsampling_cuda_launcher(int b, int n, const float *xyz, const int *offset, "
                                      "const int *new_offset, float *tmp, int *idx);",
}


def test_reference_launcher_symbols_and_prototypes():
    """The reference's own extern "C" launchers are exported under their names with their exact prototypes (so knnquery_cuda.cpp /
    sampling_cuda.cpp link against libetch_hip.so unchanged); where /root/reference is present the prototypes are also read from
    its headers."""
    import re
    B.build()
    names = _lib.declared_launchers()
    assert set(REFERENCE_LAUNCHERS) <= set(names) and len(names) == 6, names
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), n
    hdr = open(_lib.HEADER).read()
    for name, proto in REFERENCE_LAUNCHERS.items():
        mine = re.search(r"void\s+" + name + r"\s*\([^)]*\)\s*;", hdr).group(0)
        assert _norm(mine) == _norm(proto), (mine, proto)
    ref = "/root/reference/external/pointops/src"
    if os.path.isdir(ref):
        for name, rel in (("knnquery_cuda_launcher", "knnquery/knnquery_cuda_kernel.h"), ("furthestsampling_cuda_launcher", "sampling/sampling_cuda_kernel.h")):
            theirs = re.search(r"void\s+" + name + r"\s*\([^)]*\)\s*;", open(os.path.join(ref, rel)).read()).group(0)
            assert _norm(theirs) == _norm(REFERENCE_LAUNCHERS[name])
    assert "etch_knnquery_dev" in _lib.declared_symbols()
