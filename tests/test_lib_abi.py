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
