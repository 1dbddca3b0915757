"""HIP streams restricted to a subset of the compute units (hipExtStreamCreateWithCUMask), wrapped as torch streams -- the A/B switch behind VERDICT r04
item 4 ("give the Point-Transformer streams their own compute units").  Off unless ETCH_CU_PARTITION is set:

    ETCH_CU_PARTITION=<k>[:<layout>]   k = compute units per XCD reserved for the SIDE streams (the two Point-Transformer head streams, the index stream,
                                       the stage-2 stream); the stage-1 streams get the complement.  layout = how mask bit i maps to (XCD, CU):
                                       "rr" (default)  bit i -> XCD i % 8, CU i // 8   (the runtime deals the bits round-robin over the XCDs)
                                       "blk"           bit i -> XCD i // 32, CU i % 32
The measurement (profiles/r05_cu_partition_ab.txt) decides whether anything uses it; results never depend on it (scheduling only)."""
import ctypes
import os

import torch

_hip = None


def _lib():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
    return _hip


def partition():
    """-> None, or (side_mask_bits, main_mask_bits) as lists of 256 0/1 flags."""
    spec = os.environ.get("ETCH_CU_PARTITION")
    if not spec:
        return None
    k, _, layout = spec.partition(":")
    k = int(k)
    assert 0 < k < 32
    side = [0] * 256
    for i in range(256):
        cu = i // 8 if layout in ("", "rr") else i % 32
        side[i] = 1 if cu < k else 0
    return side, [1 - b for b in side]


def masked_stream(bits, priority=0):
    words = (ctypes.c_uint32 * 8)(*[sum(bits[32 * w + b] << b for b in range(32)) for w in range(8)])
    st = ctypes.c_void_p()
    rc = _lib().hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(st.value)


def make_stream(role, priority=0):
    """role: "side" (heads, index ops, stage 2) or "main" (stage 1).  A plain torch stream unless ETCH_CU_PARTITION is set."""
    part = partition()
    if part is None:
        return torch.cuda.Stream(priority=priority)
    return masked_stream(part[0] if role == "side" else part[1], priority)
