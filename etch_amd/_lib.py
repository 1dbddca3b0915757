"""ctypes loader of libetch_hip.so.  The product path has NO fallback: if the HIP library is
missing or a symbol is absent this raises, loudly."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ETCH_HIP_LIB") or os.path.join(_HERE, "lib", "libetch_hip.so")   # override: kernel-variant experiments only
HEADER = os.path.join(os.path.dirname(_HERE), "include", "etch_hip.h")
_lib = None


class EtchHipError(RuntimeError):
    pass


def has_experiments():
    """True if the loaded library was built with ETCH_BUILD_EXPERIMENTS=1 (the opt-in kernels measured slower than the default path:
    etch_inter_so3conv32, etch_pt_block_k1 / _k2)."""
    return hasattr(lib()._cdll, "etch_pt_block_k1")


def declared_symbols(experiments=False):
    """Every function declared in include/etch_hip.h; the `#ifdef ETCH_BUILD_EXPERIMENTS` sections only on request."""
    txt = open(HEADER).read()
    if not experiments:
        txt = re.sub(r"#ifdef ETCH_BUILD_EXPERIMENTS.*?#endif", "", txt, flags=re.S)
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(etch_\w+)\s*\(", txt)))


def declared_launchers():
    """The reference's own launcher symbols re-exported with their exact signatures (`void <name>_launcher(...)`)."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\bvoid\s+(\w+_launcher)\s*\(", txt)))


class _Proxy:
    """Attribute access -> C function.  When a profiler callback is installed (bench.py) every call is
    bracketed by HIP events on the current stream; otherwise the raw ctypes function is returned."""

    def __init__(self, cdll):
        self._cdll = cdll
        self.profiler = None      # callable(name, args, fn) -> status

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if self.profiler is None:
            return fn
        prof = self.profiler
        return lambda *a: prof(name, a, fn)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EtchHipError(
                f"{LIB_PATH} is missing: build it with `python -m etch_amd.build` (hipcc, gfx950). "
                "etch_amd has no CPU fallback.")
        cdll = ctypes.CDLL(LIB_PATH)
        for name in declared_symbols(experiments=hasattr(cdll, "etch_pt_block_k1")):
            fn = getattr(cdll, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = ctypes.c_int
        cdll.etch_smpl_lm_split_workspace_bytes.restype = ctypes.c_long
        for name in declared_launchers():
            getattr(cdll, name).restype = None
        _lib = _Proxy(cdll)
    return _lib


def check(status, what):
    if status != 0:
        kind = {-1: "invalid argument", -2: "unsupported size"}.get(status, f"hipError {status}")
        raise EtchHipError(f"{what} failed: {kind}")
