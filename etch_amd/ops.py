"""Torch-tensor front-end of the C ABI (include/etch_hip.h).

PyTorch is plumbing only: it owns device memory and the current HIP stream.  Every function here
hands raw device pointers to libetch_hip.so; there is no eager/CPU fallback -- a CPU tensor or a
missing library raises.
"""
import ctypes
import math
import os

import torch

from . import _lib

_c_int, _c_float, _vp = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream's handle.  Asked ~300 times per step: torch.cuda.current_stream() builds a Stream object (10 us each, 3 ms of the host's
    8.5 ms per step); the raw accessor returns the integer."""
    if _raw_stream is not None:
        return _vp(_raw_stream(torch.cuda.current_device()))
    return _vp(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return _vp(t.data_ptr())


def _need(t, dtype, name):
    if not t.is_cuda:
        raise _lib.EtchHipError(f"{name} must be a CUDA(HIP) tensor: etch_amd has no CPU path")
    if t.dtype != dtype:
        raise _lib.EtchHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.EtchHipError(f"{name} must be contiguous")  # grouping_cuda.cpp:66-68 CHECK_INPUT
    return t


# ------------------------------------------------------------------ epn_grouping / epn_gathering
def ball_query(new_xyz, xyz, radius, nsample):
    """epn_grouping.ball_query (grouping_cuda.cpp:71-86): (b,3,m),(b,3,n) -> idx (b,m,nsample) int32."""
    _need(new_xyz, torch.float32, "new_xyz"), _need(xyz, torch.float32, "xyz")
    b, _, m = new_xyz.shape
    n = xyz.shape[2]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz.device)
    _lib.check(_lib.lib().etch_ball_query(b, n, m, _c_float(radius), int(nsample), _ptr(new_xyz), _ptr(xyz), _ptr(idx), _stream()),
               "etch_ball_query")
    return idx


FPS_SPLIT_FORCE = None      # tests: an int G forces the split kernels (2..8) or disables them (0 / 1); None = fps_split_default


def fps_split_default(nseg, n_max, device=None):
    """Workgroups per scan for the FPS kernels.  Measured (profiles/r04_fps_split.txt): the per-round exchange between the workgroups of a
    split scan costs 0.75 us (one scan) to 1.2 us (32 - 64 scans in flight) on top of a fixed 0.8 us per round, against 1.1 us for the WHOLE
    one-workgroup round at 5 000 points and 2.3 us at 20 000 -- the split is slower at every size the path runs (2.7 us at 20 000 points with
    G = 4), so the default is one workgroup per scan.  The split kernels stay available (bit-identical picks): split=G, FPS_SPLIT_FORCE or
    ETCH_FPS_SPLIT=G; they need nseg * G co-resident workgroups (one per compute unit)."""
    if FPS_SPLIT_FORCE is not None:
        return int(FPS_SPLIT_FORCE) if int(FPS_SPLIT_FORCE) >= 2 else 1
    env = os.environ.get("ETCH_FPS_SPLIT")
    if env is not None and int(env) >= 2:
        cus = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
        g = min(8, int(env))
        while g >= 2 and (nseg * g > cus or n_max > 8192 * g):
            g = g // 2 if nseg * g > cus else 0
        return g if g >= 2 else 1
    return 1


def _fps_split_ws(nseg, G, device):
    ws = torch.empty((_lib.lib().etch_fps_split_workspace_bytes(nseg, G),), dtype=torch.uint8, device=device)
    return ws


def fps_split_failed(idx):
    """True if the split FPS launch that produced `idx` gave up (a workgroup of a scan never arrived); synchronises the current stream."""
    ws = getattr(idx, "_etch_split_ws", None)
    if ws is None:
        return False
    out = ctypes.c_int(0)
    _lib.check(_lib.lib().etch_fps_split_failed(_ptr(ws), ctypes.byref(out), _stream()), "etch_fps_split_failed")
    return bool(out.value)


def _fps_split_checked(idx, split):
    """A split launch that the DISPATCH chose (split=None: ETCH_FPS_SPLIT / FPS_SPLIT_FORCE -- the model path, whose callers gather with the result at
    once) is verified here: a scan that gave up carries INT_MIN indices, which must not reach a gather.  Costs a stream synchronisation
    on this opt-in path only; an explicit split=G leaves the check to the caller (fps_split_failed), as the failure-path test does."""
    if split is None and fps_split_failed(idx):
        raise _lib.EtchHipError("split FPS gave up (a workgroup of a scan never arrived): rerun with ETCH_FPS_SPLIT unset (one workgroup per scan)")
    return idx


# One-shot hand-over of sampling results that fps_pair computed ahead of their callers (models_pointcloud._prefetch_indices): key -> idx.  A result is taken
# (popped) by the first call whose arguments match; the keys hold data pointers of tensors the prefetcher keeps alive until then.
_FPS_READY = {}


def fps_pair(xyz_b3n, m, xyz_packed, offset, new_offset, new_offset_host):
    """The encoder's FPS (furthest_point_sampling(xyz_b3n, m)) and the Point-Transformer nets' first FPS level (furthestsampling(xyz_packed, offset,
    new_offset)) over the same b scans of n points each in ONE launch of 2 b workgroups (etch_fps_pair): both are one workgroup per scan and depend on the
    coordinates only.  -> (idx_a (b,m) int32, idx_b (sum m') int32), bit-identical to the separate calls."""
    _need(xyz_b3n, torch.float32, "xyz_b3n"), _need(xyz_packed, torch.float32, "xyz_packed")
    _need(offset, torch.int32, "offset"), _need(new_offset, torch.int32, "new_offset")
    b, _, n = xyz_b3n.shape
    assert xyz_packed.shape == (b * n, 3)
    idx_a = torch.zeros((b, m), dtype=torch.int32, device=xyz_b3n.device)
    idx_b = torch.zeros((int(new_offset_host[-1]),), dtype=torch.int32, device=xyz_b3n.device)
    _lib.check(_lib.lib().etch_fps_pair(b, n, int(m), _ptr(xyz_b3n), _ptr(idx_a), _ptr(xyz_packed), _ptr(offset), _ptr(new_offset), _ptr(idx_b), _stream()),
               "etch_fps_pair")
    return idx_a, idx_b


def furthest_point_sampling(xyz, m, split=None):
    """epn_grouping.furthest_point_sampling (grouping_cuda.cpp:158-173): (b,3,n) -> (b,m) int32.  split = workgroups per scan (None: auto)."""
    _need(xyz, torch.float32, "xyz")
    b, _, n = xyz.shape
    ready = _FPS_READY.pop(("vgtk", xyz.data_ptr(), int(m)), None) if _FPS_READY else None
    if ready is not None:
        return ready
    idx = torch.zeros((b, m), dtype=torch.int32, device=xyz.device)
    G = fps_split_default(b, n, xyz.device) if split is None else int(split)
    if G >= 2:
        ws = _fps_split_ws(b, G, xyz.device)
        _lib.check(_lib.lib().etch_furthest_point_sampling_split(b, n, int(m), _ptr(xyz), _ptr(idx), G, _ptr(ws), _stream()),
                   "etch_furthest_point_sampling_split")
        idx._etch_split_ws = ws
        return _fps_split_checked(idx, split)
    _lib.check(_lib.lib().etch_furthest_point_sampling(b, n, int(m), _ptr(xyz), _ptr(idx), _stream()), "etch_furthest_point_sampling")
    return idx


def gather_points_forward(points, idx):
    """epn_gathering.gather_points_forward (gathering_cuda.cpp:29-46): (b,c,n),(b,m) -> (b,c,m)."""
    _need(points, torch.float32, "points"), _need(idx, torch.int32, "idx")
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    _lib.check(_lib.lib().etch_gather_points(b, c, n, m, _ptr(points), _ptr(idx), _ptr(out), _stream()), "etch_gather_points")
    return out


def gather_points_backward(grad_out, idx, npoint):
    """epn_gathering.gather_points_backward (gathering_cuda.cpp:45-60): (b,c,m),(b,m),n -> (b,c,n)."""
    _need(grad_out, torch.float32, "grad_out"), _need(idx, torch.int32, "idx")
    b, c, m = grad_out.shape
    out = torch.empty((b, c, int(npoint)), dtype=torch.float32, device=grad_out.device)
    _lib.check(_lib.lib().etch_gather_points_backward(b, c, int(npoint), m, _ptr(grad_out), _ptr(idx), _ptr(out), _stream()),
               "etch_gather_points_backward")
    return out


# ------------------------------------------------------------------ pointops_cuda
def _seg_max(offset_host):
    prev, mx = 0, 0
    for v in offset_host:
        mx = max(mx, int(v) - prev)
        prev = int(v)
    return mx


def knnquery(nsample, xyz, new_xyz, offset, new_offset, new_offset_host=None, sqrt=True, wave_kernel=True):
    """pointops.knnquery (pointops.py:32-45): -> idx (m,nsample) int32, dist (m,nsample) = sqrt(d2).

    `new_offset_host` (list of ints) avoids a device->host sync for the grid size; if omitted it is
    read back from `new_offset` (one sync, like the reference's own .item() calls)."""
    if new_xyz is None:
        new_xyz = xyz
    _need(xyz, torch.float32, "xyz"), _need(new_xyz, torch.float32, "new_xyz")
    _need(offset, torch.int32, "offset"), _need(new_offset, torch.int32, "new_offset")
    if new_offset_host is None:
        new_offset_host = new_offset.tolist()
    m = new_xyz.shape[0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    dist = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    _lib.check(_lib.lib().etch_knnquery(len(new_offset_host), _seg_max(new_offset_host), int(m) if wave_kernel else 0, int(nsample), _ptr(xyz), _ptr(new_xyz),
                                        _ptr(offset), _ptr(new_offset), _ptr(idx), _ptr(dist), 1 if sqrt else 0, _stream()),
               "etch_knnquery")
    return idx, dist


def furthestsampling(xyz, offset, new_offset, offset_host=None, new_offset_host=None, split=None):
    """pointops.furthestsampling (pointops.py:10-28): packed (n,3) + offsets -> idx (m) int32."""
    _need(xyz, torch.float32, "xyz"), _need(offset, torch.int32, "offset"), _need(new_offset, torch.int32, "new_offset")
    if offset_host is None:
        offset_host = offset.tolist()
    if new_offset_host is None:
        new_offset_host = new_offset.tolist()
    ready = _FPS_READY.pop(("pointops", xyz.data_ptr(), new_offset.data_ptr()), None) if _FPS_READY else None
    if ready is not None:
        return ready
    idx = torch.zeros((int(new_offset_host[-1]),), dtype=torch.int32, device=xyz.device)
    nseg, n_max = len(offset_host), _seg_max(offset_host)
    G = fps_split_default(nseg, n_max, xyz.device) if split is None else int(split)
    if G >= 2:
        ws = _fps_split_ws(nseg, G, xyz.device)
        _lib.check(_lib.lib().etch_furthestsampling_split(nseg, n_max, _ptr(xyz), _ptr(offset), _ptr(new_offset), _ptr(idx), G, _ptr(ws), _stream()),
                   "etch_furthestsampling_split")
        idx._etch_split_ws = ws
        return _fps_split_checked(idx, split)
    _lib.check(_lib.lib().etch_furthestsampling(nseg, n_max, _ptr(xyz), _ptr(offset), _ptr(new_offset), _ptr(idx), _stream()), "etch_furthestsampling")
    return idx


# ------------------------------------------------------------------ dense layers
_c_long = ctypes.c_long
ACT = {None: 0, "none": 0, "relu": 1, "leaky_relu": 2}


def _optptr(t):
    return _vp(0) if t is None else _ptr(t)


def linear(x, weight, bias=None, scale=None, shift=None, act=None, res=None, res_mode=0, row_idx=None, grp=1, p_in=0, p_out=1,
           rows=None, out=None):
    """Y = epi(X[rowmap] @ W^T) via etch_linear.  x [*, K] (last dim contiguous), weight [O, K]."""
    _need(weight, torch.float32, "weight")
    if x.dtype != torch.float32 or not x.is_cuda:
        raise _lib.EtchHipError("x must be a float32 CUDA(HIP) tensor")
    K = x.shape[-1]
    assert x.stride(-1) == 1 and weight.shape[1] == K, (x.shape, weight.shape)
    x2 = x if x.dim() == 2 else x.reshape(-1, K)
    ldx = x2.stride(0) if x2.shape[0] > 1 else K
    R = x2.shape[0] if rows is None else rows
    O = weight.shape[0]
    if out is None:
        out = torch.empty((R, O), dtype=torch.float32, device=x.device)
    ldr = res.stride(0) if res is not None and res.dim() == 2 and res.shape[0] > 1 else O
    _lib.check(_lib.lib().etch_linear(int(R), int(K), int(O), _ptr(x2), _c_long(ldx), _optptr(row_idx), int(grp), int(p_in), int(p_out),
                                      _ptr(weight), _c_long(weight.stride(0)), _optptr(bias), _optptr(scale), _optptr(shift),
                                      ACT[act], _optptr(res), _c_long(ldr), int(res_mode), _ptr(out), _c_long(out.stride(0)), _stream()),
               "etch_linear")
    return out


def permute_weight_frag(w2):
    """[O, K] (O % 16 == 0, K % 16 == 0) -> MFMA fragment order Wp[t][mt][lane][s] = W2[16mt + lane%16][16t + 4(lane//16) + s]."""
    O, K = w2.shape
    assert O % 16 == 0 and K % 16 == 0
    return w2.reshape(O // 16, 16, K // 16, 4, 4).permute(2, 0, 3, 1, 4).contiguous().reshape(-1)


def permute_weight_frag32(w2):
    """[O, K] (O % 32 == 0, K % 8 == 0) -> 32x32x2 MFMA fragment order Wp[t][mt][lane][s] = W2[32mt + lane%32][8t + 4(lane//32) + s]."""
    O, K = w2.shape
    assert O % 32 == 0 and K % 8 == 0
    return w2.reshape(O // 32, 32, K // 8, 2, 4).permute(2, 0, 3, 1, 4).contiguous().reshape(-1)


def inter_weight_frag(W, cin, ks=24):
    """Fragment-ordered weight of the fused inter conv: the columns of W [cout, cin*ks] are first brought into the kernel's
    contraction order (csrc/so3conv.hip: channels are processed in passes of CCH = min(cin, 64); within a pass a lane gathers
    VEC = CCH/16 consecutive channels per load, so column r of c-tile mi is channel VEC*r + mi and sits at position 16*mi + r of its
    half; halves h of the X1 tile hold the tiles [h*MTH, (h+1)*MTH)), then permuted into MFMA fragment order."""
    return permute_weight_frag(W[:, _inter_contraction_cols(cin, ks, W.device)].contiguous())


_CONTRACTION_COLS = {}


def _inter_contraction_cols(cin, ks, device):
    """Column permutation of W [cout, cin*ks] into the contraction order of inter_so3conv_kernel (see inter_weight_frag); one upload per (shape, device)."""
    key = (cin, ks, str(device))
    if key not in _CONTRACTION_COLS:
        _CONTRACTION_COLS[key] = _inter_contraction_cols_build(cin, ks, device)
    return _CONTRACTION_COLS[key]


def _inter_contraction_cols_build(cin, ks, device):
    cch = min(cin, 64)
    vec = cch // 16
    halves = 2 if cch >= 32 else 1
    mth = vec // halves
    one = [vec * (cc % 16) + h * mth + cc // 16 for h in range(halves) for cc in range(cch // halves)]
    order = [p * cch + c for p in range(cin // cch) for c in one]
    assert sorted(order) == list(range(cin))
    return torch.tensor([c * ks + k for c in order for k in range(ks)], dtype=torch.long, device=device)


def split3_bf16(x):
    """fp32 tensor -> (3, *x.shape) int16: the bf16 bit patterns of the exact split x = hi + mid + lo (8 + 8 + 8 mantissa bits, by truncation;
    csrc/so3conv.hip split3_pack4 is the device-side twin)."""
    x = x.contiguous()
    hi = x.view(torch.int32) & -65536
    r = x - hi.view(torch.float32)
    mid = r.view(torch.int32) & -65536
    lo = (r - mid.view(torch.float32)).view(torch.int32)
    return torch.stack([(t >> 16).to(torch.int16) for t in (hi, mid, lo)])


def inter_weight_split(W, cin, ks=24, natural=False):
    """Weight of the inter conv for the split-operand kernels: columns in the kernel's contraction order (etch_inter_so3conv_split: the one of
    inter_weight_frag; natural=True, etch_inter_so3conv_planes: W's own column order c * 24 + k), split into three bf16 planes, in
    v_mfma_f32_16x16x32_bf16 A-fragment order: [chunk of 32 kappas][o tile][plane][lane = 16 * (k / 8) + o % 16][8]."""
    cout = W.shape[0]
    assert cin % 16 == 0 and cout % 16 == 0 and W.shape[1] == cin * ks and ks == 24
    planes = split3_bf16(W if natural else W[:, _inter_contraction_cols(cin, ks, W.device)])     # [3][cout][K]
    K = cin * ks
    q = planes.reshape(3, cout // 16, 16, K // 32, 4, 8)                 # [pl][mt][ol][tg][kg][e]
    return q.permute(3, 1, 0, 4, 2, 5).contiguous().reshape(-1)          # [tg][mt][pl][kg][ol][e]


def inter_weight_split32(W, cin, ks=24):
    """Weight of etch_inter_so3conv_planes32 (v_mfma_f32_32x32x16_bf16): the columns of W [cout, cin*24] in the kernel's PHYSICAL contraction order --
    per channel half h (CH = cin / 2 channels), kernel point k, 16-byte block pb, element i: channel h*CH + 4*(pb ^ sw(k)) + i with the X1 tile's
    store swizzle sw(k) = (k >> 1) & 3 (CH = 16) / k & 7 (CH = 32) -- split into three bf16 planes, in A-fragment order
    [K step of 16][o tile of 32][plane][lane = 32 * (kappa / 8 % 2) + o % 32][8]."""
    cout = W.shape[0]
    assert ks == 24 and cin in (32, 64) and cout % 32 == 0 and W.shape[1] == cin * ks
    ch = cin // 2
    cols = [(h * ch + 4 * (pb ^ (((k >> 1) & 3) if ch == 16 else (k & 7))) + i) * ks + k
            for h in range(2) for k in range(ks) for pb in range(ch // 4) for i in range(4)]
    assert sorted(cols) == list(range(cin * ks))
    planes = split3_bf16(W[:, torch.tensor(cols, dtype=torch.long, device=W.device)])     # [3][cout][K]
    K = cin * ks
    q = planes.reshape(3, cout // 32, 32, K // 16, 2, 8)                 # [pl][mt][o][s][kg][e]
    return q.permute(3, 1, 0, 4, 2, 5).contiguous().reshape(-1)          # [s][mt][pl][kg][o][e]


def intra_weight_split(w2):
    """W2 [C, 12 C] (tap-major K: W2[o, tap * C + ch]) -> the register-resident fragments of the weight-stationary intra conv
    (csrc/so3conv_ws.hip): [mt][kq][K step][plane][lane = 32 * (k / 8) + o % 32][8] with k = 3 kq C + 16 ks + 8 (lane / 32) + e."""
    C, K = w2.shape
    assert K == 12 * C and C in (32, 64)
    nks = 3 * C // 16
    planes = split3_bf16(w2)                                             # [3][C][K]
    q = planes.reshape(3, C // 32, 32, 4, nks, 2, 8)                     # [pl][mt][i][kq][ks][kg][e]
    return q.permute(1, 3, 4, 0, 5, 2, 6).contiguous().reshape(-1)       # [mt][kq][ks][pl][kg][i][e]


def _pow2_exp(m, cap=120):
    """m >= 0 (any shape) -> k (float32, integer valued) with m 2^k in [8, 16) -- the host mirror of the device's etch_scale_exp (csrc/split_bf16.h): zero,
    subnormal (below FLT_MIN) and non-finite maxima give k = 0, and k is capped at 120 so that 2^k AND 2^-k stay normal floats (an
    uncapped 4 - frexp(m) reaches 128 .. 130 for m <= 2^-124, exp2 of which is inf)."""
    ok = (m >= 1.1754943508222875e-38) & torch.isfinite(m)
    x = torch.frexp(torch.where(ok, m, torch.ones_like(m)))[1]           # m = f 2^x, f in [0.5, 1)  ->  m 2^(4 - x) in [8, 16)
    return torch.where(ok, (4 - x).clamp(max=cap), torch.zeros_like(x)).to(torch.float32)


def _row_pow2(w, cap=120):
    """w [rows, K] -> (w with every row times the power of two that puts its maximum into [8, 16), the inverse powers float32 [rows]); exact.  The fp16
    planes of an operand carry 23 bits only near that range: scaling per ROW keeps rows far below the matrix maximum at full precision (a trained
    weight matrix's rows may differ by orders of magnitude).  Rows whose maximum is zero / subnormal / non-finite keep the factor 1 (_pow2_exp)."""
    wd = w.detach()
    e = _pow2_exp(wd.abs().amax(1), cap)
    return wd * torch.exp2(e)[:, None], torch.exp2(-e).contiguous()


def intra_weight_split_f16(w2):
    """W2 [C, 12 C] -> the fragments of etch_intra_so3conv_f16: intra_weight_split's order with TWO fp16 planes (h = fp16(w), l = fp16(w - h)) of W2 with every
    row (output channel) times its own power of two (_row_pow2; exact); the inverse powers ride on the tensor as `.wsc` (the kernel's epilogue applies them)."""
    C, K = w2.shape
    assert K == 12 * C and C in (32, 64)
    nks = 3 * C // 16
    ws, wsc = _row_pow2(w2)
    hi = ws.to(torch.float16)
    planes = torch.stack([hi, (ws - hi.float()).to(torch.float16)])      # [2][C][K]
    q = planes.reshape(2, C // 32, 32, 4, nks, 2, 8)                     # [pl][mt][i][kq][ks][kg][e]
    out = q.permute(1, 3, 4, 0, 5, 2, 6).contiguous().reshape(-1).view(torch.int16)       # [mt][kq][ks][pl][kg][i][e]
    out.wsc = wsc
    return out


def inter_weight_frag32(W, cin, ks=24):
    """Fragment order of the 32x32x2 inter conv (csrc/so3conv32.hip): [slice = 3 h + g][mt][kp][u][lane][4] with
    [lane][s] = W[32 mt + lane % 32][(32 h + c) * 24 + 8 g + 4 (lane / 32) + s], c = kp * NU + u -- h = 32-channel tile of the input,
    g = group of 8 kernel points, the slice's 256 contraction entries split over NKP = 8 / MT waves of NU = 32 / NKP k-steps."""
    cout = W.shape[0]
    assert ks == 24 and cin % 32 == 0 and cout in (32, 64) and W.shape[1] == cin * ks
    mt, ntil = cout // 32, cin // 32
    nkp = 8 // mt
    nu = 32 // nkp
    w = W.reshape(mt, 32, ntil, nkp, nu, 3, 2, 4)          # [mt][j][h][kp][u][g][kk][s]
    return w.permute(2, 5, 0, 3, 4, 6, 1, 7).contiguous().reshape(-1)


INTER_SPLIT = os.environ.get("ETCH_INTER_SPLIT", "1") != "0"      # step 2 of the inter conv on the bf16 matrix cores (split fp32 operands); 0: fp32 MFMA
INTER_MFMA32 = os.environ.get("ETCH_INTER_MFMA32", "0") == "1"     # (32|64) -> (32|64) channels: the 32x32x2 MFMA form, two points per workgroup
INTER_MFMA32_SHAPES = ((32, 32), (32, 64), (64, 64))
INTER_X = os.environ.get("ETCH_INTER_X", "1") != "0"             # both contractions of the inter conv on the bf16 matrix cores, gathered rows as bf16 planes (csrc/so3conv_x.hip)
INTER_X32 = os.environ.get("ETCH_INTER_X", "1") != "16"          # 64 input channels: the 32x32x16 MFMA form (ETCH_INTER_X=16: the 16x16x32 form everywhere)


INTER_KQ = os.environ.get("ETCH_INTER_KQ", "1") != "0"          # kernel weights formed on the matrix cores (csrc/so3conv_y.hip); 0: the round-3 kernels (fp32 MFMA step 1,
                                                                 # exact three-plane step 2) -- or, in an ETCH_BUILD_EXPERIMENTS library, the retired round-4 planes kernels


def inter_planes_form(cin):
    """MFMA shape of the round-4 planes kernel for this width: 32 (etch_inter_so3conv_planes32, 64 input channels) or 16 (etch_inter_so3conv_planes)."""
    return 32 if (INTER_X32 and cin == 64) else 16


def split2_planes_f16(x_cl):
    """x (..., C) fp32 -> (..., 2, C) float16: h = fp16(x), l = fp16(x - h), both to nearest, the layout etch_inter_so3conv_planes_kq gathers (the
    encoder's own producer, instnorm_act_add(want_planes="f16"), writes it directly)."""
    _need(x_cl, torch.float32, "x")
    C = x_cl.shape[-1]
    planes = torch.empty(tuple(x_cl.shape[:-1]) + (2, C), dtype=torch.float16, device=x_cl.device)
    _lib.check(_lib.lib().etch_split2_planes_f16(_c_long(x_cl.numel() // C), int(C), _ptr(x_cl), _ptr(planes), _stream()), "etch_split2_planes_f16")
    return planes


def split2_planes_f16_scaled(x_cl):
    """x (b, ..., C) fp32 of ANY scale -> ((b, ..., 2, C) float16 planes of x[s] * 2^k(s), fsc (b,) float32 = 2^-k(s)): every scan is brought to a maximum
    magnitude in [8, 16) before the split (exact), fsc is the factor etch_inter_so3conv_planes_kq's epilogue multiplies that scan's outputs with."""
    _need(x_cl, torch.float32, "x")
    b, C = x_cl.shape[0], x_cl.shape[-1]
    planes = torch.empty(tuple(x_cl.shape[:-1]) + (2, C), dtype=torch.float16, device=x_cl.device)
    mx = torch.zeros((b,), dtype=torch.int32, device=x_cl.device)
    fsc = torch.empty((b,), dtype=torch.float32, device=x_cl.device)
    _lib.check(_lib.lib().etch_split2_planes_f16_scaled(int(b), _c_long(x_cl.numel() // (b * C)), int(C), _ptr(x_cl), _ptr(mx), _ptr(planes), _ptr(fsc), _stream()),
               "etch_split2_planes_f16_scaled")
    return planes, fsc


def inter_weight_split32_f16(W, cin, ks=24):
    """Weight of etch_inter_so3conv_planes_kq: the columns of W [cout, cin*24] in the physical contraction order of inter_weight_split32, every row (output
    channel) times its own power of two (_row_pow2: exact; `.wsc` = the inverse powers, applied by the kernel's epilogue), as two fp16 planes
    h = fp16(w), l = fp16(w - h), in A-fragment order [K step of 16][o tile of 32][plane][lane = 32 * (kappa / 8 % 2) + o % 32][8]."""
    cout = W.shape[0]
    assert ks == 24 and cin in (32, 64) and cout % 32 == 0 and W.shape[1] == cin * ks
    ch = cin // 2
    cols = [(h * ch + 4 * (pb ^ (((k >> 1) & 3) if ch == 16 else (k & 7))) + i) * ks + k
            for h in range(2) for k in range(ks) for pb in range(ch // 4) for i in range(4)]
    assert sorted(cols) == list(range(cin * ks))
    ws, wsc = _row_pow2(W[:, torch.tensor(cols, dtype=torch.long, device=W.device)])
    hi = ws.to(torch.float16)
    planes = torch.stack([hi, (ws - hi.float()).to(torch.float16)])                       # [2][cout][K]
    K = cin * ks
    q = planes.reshape(2, cout // 32, 32, K // 16, 2, 8)                 # [pl][mt][o][s][kg][e]
    out = q.permute(3, 1, 0, 4, 2, 5).contiguous().reshape(-1)           # [s][mt][pl][kg][o][e]
    out.wsc = wsc
    return out


def inter_kpoint_operand(rk, sigma):
    """rk (60,24,3) fp32 rotated kernel points -> (60,2,64,8) int16: the kernel-point factor of the weights' pre-activation as matrix-core B
    fragments (etch_inter_so3conv_planes_kq)."""
    _need(rk, torch.float32, "rk")
    assert tuple(rk.shape) == (60, 24, 3)
    kq = torch.empty((60, 2, 64, 8), dtype=torch.int16, device=rk.device)
    _lib.check(_lib.lib().etch_inter_kpoint_operand(_c_float(sigma), _ptr(rk), _ptr(kq), _stream()), "etch_inter_kpoint_operand")
    return kq


def inter_planes_supported(cin, cout, nn):
    """Shapes etch_inter_so3conv_planes covers (the three convs of the released encoder depth)."""
    return INTER_X and (cin, cout) in ((32, 32), (32, 64), (64, 64)) and nn in (32, 64)      # = etch_inter_so3conv_planes_supported


def split3_planes(x_cl):
    """x (..., C) fp32 -> (..., 3, C) int16: the bf16 bit patterns of the exact split x = hi + mid + lo, the layout etch_inter_so3conv_planes
    gathers (the encoder's own producer, instnorm_act_add(want_planes=True), writes it directly)."""
    _need(x_cl, torch.float32, "x")
    C = x_cl.shape[-1]
    planes = torch.empty(tuple(x_cl.shape[:-1]) + (3, C), dtype=torch.int16, device=x_cl.device)
    _lib.check(_lib.lib().etch_split3_planes(_c_long(x_cl.numel() // C), int(C), _ptr(x_cl), _ptr(planes), _stream()), "etch_split3_planes")
    return planes


# ------------------------------------------------------------------ EPN encoder
def spatial_order(xyz):
    """xyz (b,3,n) -> (b,n) int32 Morton order of each scan (scheduling hint for inter_so3conv / prop_interp: a permutation of
    0..n-1 per scan; 30-bit codes up to 16 384 points, 15-bit codes in 32 768-point slices above)."""
    _need(xyz, torch.float32, "xyz")
    b, _, n = xyz.shape
    order = torch.empty((b, n), dtype=torch.int32, device=xyz.device)
    _lib.check(_lib.lib().etch_spatial_order(b, n, _ptr(xyz), _ptr(order), _stream()), "etch_spatial_order")
    return order


def inter_so3conv(xyz, new_xyz, ball_idx, feats_cl, rk, W, Wp, bias, sigma, order=None, want_stats=False, Wp32=None, Wq=None, Wqn=None,
                  feats_planes=None, Wq32=None, kq=None, Wqh=None):
    """feats_cl (b,p1,60,cin) channels-last -> (b,p2,60,cout) pre-norm.  order (b,p2) int32: processing order of the output points.
    want_stats: also return the InstanceNorm (mean, rstd) of the output, accumulated in the conv's epilogue.
    kq (inter_kpoint_operand) + Wqh (inter_weight_split32_f16): the round-5 kernel (weights' pre-activation on the matrix cores, two fp16 planes per
    operand, every covered shape); feats_planes (b,p1,60,2,cin) float16 = split2_planes_f16 (made here when absent or in the bf16 format).
    Wq32 (inter_weight_split32) / Wqn (inter_weight_split(natural=True)): both contractions on the bf16 matrix cores where the shape is covered
    (32x32x16 for 64 input channels / 16x16x32 MFMA form); feats_planes (b,p1,60,3,cin) int16 = the producer's split of feats_cl (made here when absent)."""
    b, p1, na, cin = feats_cl.shape
    p2, nn = ball_idx.shape[1], ball_idx.shape[2]
    cout = W.shape[0]
    for t, n in ((xyz, "xyz"), (new_xyz, "new_xyz"), (feats_cl, "feats"), (rk, "rk"), (W, "W"), (bias, "bias")):
        _need(t, torch.float32, n)
    _need(ball_idx, torch.int32, "ball_idx")
    if order is not None:
        _need(order, torch.int32, "order")
        assert tuple(order.shape) == (b, p2)
    out = torch.empty((b, p2, 60, cout), dtype=torch.float32, device=xyz.device)
    fused = want_stats and 256 % cout == 0 and (cin >= 16 or (cin == 1 and cout <= 64 and nn * 60 >= 1026))      # the c1 kernel's own precondition (so3conv.hip)
    part = torch.empty((b, p2, 2, cout), dtype=torch.float64, device=xyz.device) if fused else None
    if kq is not None and Wqh is not None and INTER_KQ and inter_planes_supported(cin, cout, nn):
        _need(Wqh, torch.float16, "Wqh"), _need(kq, torch.int16, "kq")
        fsc = None
        if feats_planes is None or feats_planes.dtype != torch.float16:
            # the caller's own features, of unknown scale (the operator API has no domain restriction): planes of every scan times its own power of two,
            # undone by the kernel's epilogue (fsc).  Planes that arrive with the features are the producer's (instnorm_act_add: unit scale by construction)
            feats_planes, fsc = split2_planes_f16_scaled(feats_cl)
        _need(feats_planes, torch.float16, "feats_planes")
        assert tuple(feats_planes.shape) == (b, p1, na, 2, cin) and kq.numel() == 60 * 2 * 64 * 8 and Wqh.numel() == 2 * cout * cin * 24
        _lib.check(_lib.lib().etch_inter_so3conv_planes_kq(b, cin, cout, p1, p2, nn, _c_float(sigma), _ptr(xyz), _ptr(new_xyz), _ptr(ball_idx),
                                                          _ptr(feats_planes), _ptr(kq), _ptr(Wqh), _ptr(Wqh.wsc), _optptr(fsc), _ptr(bias), _ptr(out), _optptr(order),
                                                          _optptr(part), _stream()), "etch_inter_so3conv_planes_kq")
    elif Wq32 is not None and cin == 64 and inter_planes_form(cin) == 32 and inter_planes_supported(cin, cout, nn) and _lib.has_experiments():
        _need(Wq32, torch.int16, "Wq32")
        if feats_planes is None or feats_planes.dtype != torch.int16:
            feats_planes = split3_planes(feats_cl)
        _need(feats_planes, torch.int16, "feats_planes")
        assert tuple(feats_planes.shape) == (b, p1, na, 3, cin)
        _lib.check(_lib.lib().etch_inter_so3conv_planes32(b, cin, cout, p1, p2, nn, _c_float(sigma), _ptr(xyz), _ptr(new_xyz), _ptr(ball_idx),
                                                         _ptr(feats_planes), _ptr(rk), _ptr(Wq32), _ptr(bias), _ptr(out), _optptr(order),
                                                         _optptr(part), _stream()), "etch_inter_so3conv_planes32")
    elif Wqn is not None and inter_planes_supported(cin, cout, nn) and _lib.has_experiments():
        _need(Wqn, torch.int16, "Wqn")
        if feats_planes is None or feats_planes.dtype != torch.int16:
            feats_planes = split3_planes(feats_cl)
        _need(feats_planes, torch.int16, "feats_planes")
        assert tuple(feats_planes.shape) == (b, p1, na, 3, cin)
        _lib.check(_lib.lib().etch_inter_so3conv_planes(b, cin, cout, p1, p2, nn, _c_float(sigma), _ptr(xyz), _ptr(new_xyz), _ptr(ball_idx),
                                                       _ptr(feats_planes), _ptr(rk), _ptr(Wqn), _ptr(bias), _ptr(out), _optptr(order),
                                                       _optptr(part), _stream()), "etch_inter_so3conv_planes")
    elif Wp32 is not None and INTER_MFMA32 and (cin, cout) in INTER_MFMA32_SHAPES and _lib.has_experiments():
        _need(Wp32, torch.float32, "Wp32")
        _lib.check(_lib.lib().etch_inter_so3conv32(b, cin, cout, p1, p2, nn, _c_float(sigma), _ptr(xyz), _ptr(new_xyz), _ptr(ball_idx),
                                                   _ptr(feats_cl), _ptr(rk), _ptr(Wp32), _ptr(bias), _ptr(out), _optptr(order),
                                                   _optptr(part), _stream()), "etch_inter_so3conv32")
    elif Wq is not None and INTER_SPLIT and cin % 16 == 0:
        _need(Wq, torch.int16, "Wq")
        _lib.check(_lib.lib().etch_inter_so3conv_split(b, cin, cout, p1, p2, nn, _c_float(sigma), _ptr(xyz), _ptr(new_xyz), _ptr(ball_idx),
                                                      _ptr(feats_cl), _ptr(rk), _ptr(Wq), _ptr(bias), _ptr(out), _optptr(order),
                                                      _optptr(part), _stream()), "etch_inter_so3conv_split")
    else:
        # all-ones features (the encoder's occupancy tensor, tagged by models/so3conv.py): the one-channel kernel takes NULL and skips the feature gather
        unit = cin == 1 and getattr(feats_cl, "_etch_constant", None) == 1.0 and cout <= 64 and nn * 60 >= 1026
        _lib.check(_lib.lib().etch_inter_so3conv_ordered(b, cin, cout, p1, p2, nn, _c_float(sigma), _ptr(xyz), _ptr(new_xyz), _ptr(ball_idx),
                                                         _vp(0) if unit else _ptr(feats_cl), _ptr(rk), _ptr(W), _optptr(Wp), _ptr(bias), _ptr(out), _optptr(order),
                                                         _optptr(part), _stream()), "etch_inter_so3conv")
    if not want_stats:
        return out
    if not fused:
        return out, instnorm_stats(out)
    mean = torch.empty((b, cout), dtype=torch.float32, device=xyz.device)
    rstd = torch.empty((b, cout), dtype=torch.float32, device=xyz.device)
    _lib.check(_lib.lib().etch_instnorm_from_partials(b, p2, cout, 60, _ptr(part), _ptr(mean), _ptr(rstd), _stream()), "etch_instnorm_from_partials")
    return out, (mean, rstd)


INTRA_MFMA32 = os.environ.get("ETCH_INTRA_MFMA32", "1") != "0"     # widths 32 / 64: the 32x32x2 MFMA form (ETCH_INTRA_MFMA32=0: the 16x16x4 kernel)
INTRA_SPLIT = os.environ.get("ETCH_INTRA_SPLIT", "1") != "0"       # widths 32 / 64: weight-stationary on the bf16 matrix cores, split fp32 operands
INTRA_F16 = os.environ.get("ETCH_INTRA_SPLIT", "1") != "bf16"     # ... as two fp16 planes and three cross terms (round 5); "bf16": the three-plane bf16 form


def intra_so3conv(x_cl, intra_idx32, Wp, bias, cout, mean=None, rstd=None, want_stats=False, Wp32=None, Wq=None, Wqh=None):
    """want_stats: also return the InstanceNorm (mean, rstd) of the output, accumulated in the conv's epilogue (even point counts;
    otherwise by the separate statistics pass)."""
    b, p, na, c = x_cl.shape
    _need(x_cl, torch.float32, "x"), _need(intra_idx32, torch.int32, "intra_idx"), _need(Wp, torch.float32, "Wp")
    out = torch.empty((b, p, 60, cout), dtype=torch.float32, device=x_cl.device)
    fused = want_stats and p % 2 == 0 and c <= 64        # wider tiles (encoder depths 3 / 4) take the separate statistics pass
    part = torch.empty((b * (p // 2), 2, cout), dtype=torch.float64, device=x_cl.device) if fused else None
    # the two-plane fp16 form splits the NORMALISED tile (unit scale by construction); without statistics the rows are the caller's own, of unknown scale
    # (> 65 504 -> inf, far below 1 -> the planes' absolute floor): those calls take the exact forms below
    if Wqh is not None and mean is not None and rstd is not None and INTRA_SPLIT and INTRA_F16 and c == cout and c in (32, 64) and (part is None or p % 2 == 0):
        _need(Wqh, torch.int16, "Wqh")
        _lib.check(_lib.lib().etch_intra_so3conv_f16(b, c, cout, p, _ptr(x_cl), _optptr(mean), _optptr(rstd), _ptr(intra_idx32), _ptr(Wqh),
                                                    _ptr(Wqh.wsc), _ptr(bias), _ptr(out), _optptr(part), _stream()), "etch_intra_so3conv_f16")
    elif Wq is not None and INTRA_SPLIT and c == cout and c in (32, 64) and (part is None or p % 2 == 0):
        _need(Wq, torch.int16, "Wq")
        _lib.check(_lib.lib().etch_intra_so3conv_split(b, c, cout, p, _ptr(x_cl), _optptr(mean), _optptr(rstd), _ptr(intra_idx32), _ptr(Wq),
                                                      _ptr(bias), _ptr(out), _optptr(part), _stream()), "etch_intra_so3conv_split")
    elif Wp32 is not None and INTRA_MFMA32 and c == cout and c in (32, 64):
        _lib.check(_lib.lib().etch_intra_so3conv32(b, c, cout, p, _ptr(x_cl), _optptr(mean), _optptr(rstd), _ptr(intra_idx32), _ptr(Wp32),
                                                   _ptr(bias), _ptr(out), _optptr(part), _stream()), "etch_intra_so3conv32")
    else:
        _lib.check(_lib.lib().etch_intra_so3conv_stats(b, c, cout, p, _ptr(x_cl), _optptr(mean), _optptr(rstd), _ptr(intra_idx32), _ptr(Wp),
                                                       _ptr(bias), _ptr(out), _optptr(part), _stream()), "etch_intra_so3conv")
    if not want_stats:
        return out
    if not fused:
        return out, instnorm_stats(out)
    m = torch.empty((b, cout), dtype=torch.float32, device=x_cl.device)
    r = torch.empty((b, cout), dtype=torch.float32, device=x_cl.device)
    _lib.check(_lib.lib().etch_instnorm_from_partials(b, p // 2, cout, 120, _ptr(part), _ptr(m), _ptr(r), _stream()), "etch_instnorm_from_partials")
    return out, (m, r)


def instnorm_stats(x_cl):
    """x (b, ..., C) -> mean (b,C), rstd (b,C) over all middle dims."""
    _need(x_cl, torch.float32, "x")
    b, C = x_cl.shape[0], x_cl.shape[-1]
    rows = x_cl.numel() // (b * C)
    ws = torch.empty((_lib.lib().etch_instnorm_stats_workspace_bytes(b, C) // 8,), dtype=torch.float64, device=x_cl.device)
    mean = torch.empty((b, C), dtype=torch.float32, device=x_cl.device)
    rstd = torch.empty((b, C), dtype=torch.float32, device=x_cl.device)
    _lib.check(_lib.lib().etch_instnorm_stats(b, rows, C, _ptr(x_cl), _ptr(ws), _ptr(mean), _ptr(rstd), _stream()), "etch_instnorm_stats")
    return mean, rstd


def instnorm_act_add_k1(x1, m1, r1, f, slope, offset, want_planes=False):
    """out = lrelu((x1 - m1) r1) + lrelu(f slope + offset): the second branch a one-channel 1x1 conv + InstanceNorm folded into per-(scan, channel)
    slope / offset (etch_instnorm_act_add_k1_planes_f16).  x1 (b, ..., C), f (b, rows) one value per row of x1, slope / offset (b, C).
    want_planes: False or "f16"."""
    _need(x1, torch.float32, "x1"), _need(f, torch.float32, "f"), _need(slope, torch.float32, "slope"), _need(offset, torch.float32, "offset")
    b, C = x1.shape[0], x1.shape[-1]
    rows = x1.numel() // (b * C)
    assert f.numel() == b * rows and tuple(slope.shape) == (b, C) and tuple(offset.shape) == (b, C) and want_planes in (False, None, "f16")
    out = torch.empty_like(x1)
    planes = torch.empty(tuple(x1.shape[:-1]) + (2, C), dtype=torch.float16, device=x1.device) if want_planes else None
    _lib.check(_lib.lib().etch_instnorm_act_add_k1_planes_f16(b, rows, C, _ptr(x1), _ptr(m1), _ptr(r1), _ptr(f.contiguous()), _ptr(slope.contiguous()),
                                                               _ptr(offset.contiguous()), _ptr(out), _optptr(planes), _stream()), "etch_instnorm_act_add_k1")
    return (out, planes) if want_planes else out


def instnorm_act_add(x1, m1, r1, x2=None, m2=None, r2=None, want_planes=False):
    """want_planes: also return the result split for the next conv's gathers -- True / "bf16": three bf16 planes (..., 3, C) int16
    (etch_inter_so3conv_planes); "f16": two fp16 planes (..., 2, C) float16 (etch_inter_so3conv_planes_kq)."""
    _need(x1, torch.float32, "x1")
    b, C = x1.shape[0], x1.shape[-1]
    rows = x1.numel() // (b * C)
    out = torch.empty_like(x1)
    f16 = want_planes == "f16"
    planes = None
    if want_planes:
        planes = torch.empty(tuple(x1.shape[:-1]) + ((2, C) if f16 else (3, C)), dtype=torch.float16 if f16 else torch.int16, device=x1.device)
    fn = _lib.lib().etch_instnorm_act_add_planes_f16 if f16 else _lib.lib().etch_instnorm_act_add_planes
    _lib.check(fn(b, rows, C, _ptr(x1), _ptr(m1), _ptr(r1), _optptr(x2), _optptr(m2), _optptr(r2), _ptr(out), _optptr(planes), _stream()),
               "etch_instnorm_act_add")
    return (out, planes) if want_planes else out


# ------------------------------------------------------------------ propagation + direction head
def prop3nn(xyz1_bn3, xyz2_b3s):
    _need(xyz1_bn3, torch.float32, "xyz1"), _need(xyz2_b3s, torch.float32, "xyz2")
    B, N, _ = xyz1_bn3.shape
    S = xyz2_b3s.shape[2]
    idx = torch.empty((B, N, 3), dtype=torch.int32, device=xyz1_bn3.device)
    w = torch.empty((B, N, 3), dtype=torch.float32, device=xyz1_bn3.device)
    _lib.check(_lib.lib().etch_prop3nn(B, N, S, _ptr(xyz1_bn3), _ptr(xyz2_b3s), _ptr(idx), _ptr(w), _stream()), "etch_prop3nn")
    return idx, w


def prop_interp(feats_cl, idx, w, order=None):
    """feats_cl (B,S,A,C) -> out (B,N,A,C), inv (B,N,C).  order (B,N) int32: processing order of the fine points (scheduling only)."""
    _need(feats_cl, torch.float32, "feats"), _need(idx, torch.int32, "idx"), _need(w, torch.float32, "w")
    if order is not None:
        _need(order, torch.int32, "order")
    B, S, A, C = feats_cl.shape
    N = idx.shape[1]
    out = torch.empty((B, N, A, C), dtype=torch.float32, device=feats_cl.device)
    inv = torch.empty((B, N, C), dtype=torch.float32, device=feats_cl.device)
    _lib.check(_lib.lib().etch_prop_interp_ordered(B, N, S, A, C, _ptr(feats_cl), _ptr(idx), _ptr(w), _ptr(out), _ptr(inv), _optptr(order),
                                                   _stream()), "etch_prop_interp")
    return out, inv


def mhsa_attention(qkv, T, qoff, koff, voff, embedding_dim=64):
    """qkv [T*60, ld] -> [T*60, embedding_dim]: 8-head attention over each point's 60 tokens (heads of embedding_dim / 8)."""
    _need(qkv, torch.float32, "qkv")
    E = int(embedding_dim)
    out = torch.empty((T * 60, E), dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().etch_mhsa_attention_dim(_c_long(T), E, _ptr(qkv), _c_long(qkv.stride(0)), qoff, koff, voff, _ptr(out), _c_long(E), _stream()),
               "etch_mhsa_attention_dim")
    return out


def mhsa_layer(x, wq, wk, wv, wc=None, bc=None, mode=0):
    """x [T*60, 64] tokens -> [T*60, 64]: one fused MultiHeadAttention layer (mode 0 residual, 1 plain, 2 heads only)."""
    for t, n in ((x, "x"), (wq, "wq"), (wk, "wk"), (wv, "wv")) + (((wc, "wc"), (bc, "bc")) if mode != 2 else ()):
        _need(t, torch.float32, n)
    assert x.shape[-1] == 64 and x.numel() % (60 * 64) == 0 and wq.shape == (64, 64)
    T = x.numel() // (60 * 64)
    out = torch.empty((T * 60, 64), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_mhsa_layer(_c_long(T), _ptr(x), _ptr(wq), _ptr(wk), _ptr(wv), _optptr(wc), _optptr(bc), int(mode), _ptr(out),
                                          _stream()), "etch_mhsa_layer")
    return out


def dirtail_weight_split(Wf):
    """Wf (128, 64) fp32 (direction_predictor.net[0] o head_combine) -> the two fp16 planes of Wf, every row (hidden unit) times its own power of two 2^kw
    (_row_pow2), as A fragments of v_mfma_f32_32x32x16_f16: [wave w = hidden tile of 32][K step ks][plane][lane = 32 * (k / 8 % 2) + h % 32][8] with
    k = 16 ks + 8 (lane / 32) + e (etch_mhsa_layer_dirtail).  `.wsc` = 2^-kw: the caller folds it into the tail's constants -- relu(2^-kw a + b) v =
    relu(a + 2^kw b) (2^-kw v) -- so the kernel needs no per-unit factor (dirtail_constants)."""
    assert tuple(Wf.shape) == (128, 64)
    ws, wsc = _row_pow2(Wf.float(), cap=60)         # the power is folded into the bias (bf 2^kw, dirtail_constants): capped so that fold cannot overflow
    hi = ws.to(torch.float16)
    planes = torch.stack([hi, (ws - hi.float()).to(torch.float16)])        # [2][128][64]
    q = planes.reshape(2, 4, 32, 4, 2, 8)                                  # [pl][w][h][ks][kg][e]
    out = q.permute(1, 3, 0, 4, 2, 5).contiguous().reshape(-1)             # [w][ks][pl][kg][h][e]
    out.wsc = wsc
    return out


def dirtail_constants(bf, v, c, wsc):
    """[bf (128) | v (128) | c] of etch_mhsa_layer_dirtail with the hidden units' powers of two folded in (exact): bf / wsc, v * wsc."""
    return torch.cat([bf / wsc, v * wsc, c.view(1)]).contiguous()


def mhsa_layer_dirtail(x, wq, wk, wv, Wfq, tab):
    """x [T*60, 64] tokens -> anc_w [T, 60]: the last MultiHeadAttention layer's heads + the folded direction tail in one kernel."""
    for t, n in ((x, "x"), (wq, "wq"), (wk, "wk"), (wv, "wv"), (tab, "tab")):
        _need(t, torch.float32, n)
    _need(Wfq, torch.float16, "Wfq")
    assert x.shape[-1] == 64 and x.numel() % (60 * 64) == 0 and wq.shape == (64, 64) and Wfq.numel() == 2 * 128 * 64 and tab.numel() == 257
    T = x.numel() // (60 * 64)
    out = torch.empty((T, 60), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_mhsa_layer_dirtail(_c_long(T), _ptr(x), _ptr(wq), _ptr(wk), _ptr(wv), _ptr(Wfq), _ptr(tab), _ptr(out), _stream()),
               "etch_mhsa_layer_dirtail")
    return out


def mhsa_interp_layer(feats_cl, idx, w, wq, wk, wv, wc, bc, order=None):
    """First (residual) MHSA layer on the 3-NN interpolated tokens, which are formed inside the kernel: feats_cl (B,S,60,64) coarse
    tokens, idx / w (B,N,3) from prop3nn -> (B*N*60, 64).  order (B,N) int32: processing order of the fine points (scheduling only)."""
    for t, n in ((feats_cl, "feats"), (w, "w"), (wq, "wq"), (wk, "wk"), (wv, "wv"), (wc, "wc")):
        _need(t, torch.float32, n)
    _need(idx, torch.int32, "idx")
    if order is not None:
        _need(order, torch.int32, "order")
    if bc is not None:
        _need(bc, torch.float32, "bc")
    B, S, A, C = feats_cl.shape
    assert (A, C) == (60, 64) and idx.shape[0] == B and idx.shape == w.shape
    N = idx.shape[1]
    out = torch.empty((B * N * 60, 64), dtype=torch.float32, device=feats_cl.device)
    sched = torch.empty((B * N, 8), dtype=torch.int32, device=feats_cl.device)      # workspace: per-slot records of the processing order
    _lib.check(_lib.lib().etch_mhsa_interp_layer(B, N, S, _ptr(feats_cl), _ptr(idx), _ptr(w), _optptr(order), _ptr(wq), _ptr(wk), _ptr(wv), _ptr(wc),
                                                 _optptr(bc), _ptr(out), _ptr(sched), _stream()), "etch_mhsa_interp_layer")
    return out


def token_mean(x):
    """x (T, A, C) -> mean over the A tokens (T, C)."""
    _need(x, torch.float32, "x")
    T, A, C = x.shape
    out = torch.empty((T, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_token_mean(_c_long(T), A, C, _ptr(x), _ptr(out), _stream()), "etch_token_mean")
    return out


def rowdot(x, w, bias):
    _need(x, torch.float32, "x"), _need(w, torch.float32, "w")
    R, K = x.shape
    y = torch.empty((R,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_rowdot(_c_long(R), K, _ptr(x), _c_long(x.stride(0)), _ptr(w), _c_float(bias), _ptr(y), _stream()), "etch_rowdot")
    return y


def so3_mean_dir(w, anchors, want_R=False, want_sv=False):
    _need(w, torch.float32, "w"), _need(anchors, torch.float32, "anchors")
    T, A = w.shape
    d = torch.empty((T, 3), dtype=torch.float32, device=w.device)
    R = torch.empty((T, 3, 3), dtype=torch.float32, device=w.device) if want_R else None
    sv = torch.empty((T, 3), dtype=torch.float32, device=w.device) if want_sv else None
    _lib.check(_lib.lib().etch_so3_mean_dir(_c_long(T), A, _ptr(w), _ptr(anchors), _ptr(d), _optptr(R), _optptr(sv), _stream()), "etch_so3_mean_dir")
    return d, R, sv


# ------------------------------------------------------------------ Point-Transformer pieces
def pt_attention(p, qkv, c, idx, params, ns):
    """qkv [n,3c] (q|k|v) ; params: list of 16 tensors-or-None -> out [n,c]."""
    n = p.shape[0]
    out = torch.empty((n, c), dtype=torch.float32, device=p.device)
    arr = (ctypes.c_void_p * 16)(*[(0 if t is None else t.data_ptr()) for t in params])
    base = qkv.data_ptr()
    _lib.check(_lib.lib().etch_pt_attention(n, c, ns, _ptr(p), _vp(base), _vp(base + 4 * c), _vp(base + 8 * c), _c_long(qkv.stride(0)),
                                            _ptr(idx), arr, _ptr(out), _c_long(c), _stream()), "etch_pt_attention")
    return out


PT_MFMA_SHAPES = {(64, 8), (128, 8), (64, 16), (128, 16), (256, 16), (512, 16)}     # (c, nsample) instantiated in pt.hip


def pt_attention_mfma(p, qkv, c, idx, params, ns, w2):
    """Same result as pt_attention: gather, linear_w MLP (matrix cores), neighbour softmax and aggregation in one kernel."""
    n = p.shape[0]
    _need(w2, torch.float32, "w2")
    out = torch.empty((n, c), dtype=torch.float32, device=p.device)
    arr = (ctypes.c_void_p * 16)(*[(0 if t is None else t.data_ptr()) for t in params])
    base = qkv.data_ptr()
    _lib.check(_lib.lib().etch_pt_attention_mfma(n, c, ns, _ptr(p), _vp(base), _vp(base + 4 * c), _vp(base + 8 * c), _c_long(qkv.stride(0)),
                                                 _ptr(idx), arr, _ptr(w2), _ptr(out), _c_long(c), _stream()), "etch_pt_attention_mfma")
    return out


def _need_experiments(what):
    if not _lib.has_experiments():
        raise _lib.EtchHipError(f"{what} is an opt-in experiment: rebuild with ETCH_BUILD_EXPERIMENTS=1 python -m etch_amd.build")


def pt_block_k1(x, w1, s1, t1, wqkv, bqkv):
    """First half of a PointTransformerBlock: qkv (n, 3c) = relu(bn1(x W1^T)) Wqkv^T + bqkv (pointtransformer_seg.py:112-113, 27)."""
    _need_experiments("etch_pt_block_k1")
    n, c = x.shape
    assert x.stride(1) == 1 and w1.shape == (c, c) and wqkv.shape == (3 * c, c)
    qkv = torch.empty((n, 3 * c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_pt_block_k1(n, c, _ptr(x), _c_long(x.stride(0)), _ptr(w1), _ptr(s1), _ptr(t1), _ptr(wqkv), _ptr(bqkv), _ptr(qkv),
                                           _c_long(3 * c), _stream()), "etch_pt_block_k1")
    return qkv


PT_BLOCK_SHAPES = {(64, 8), (128, 8), (128, 16), (256, 16), (512, 16)}     # (c, nsample) the fused block kernels are instantiated for


def pt_block_k2(p, qkv, c, idx, params, ns, w2, w3, s3, t3, x_res, next_k1=None):
    """Second half of a PointTransformerBlock: vector attention -> bn2 -> ReLU -> linear3 -> bn3 -> + x -> ReLU (pointtransformer_seg.py:
    114-121), the attention output kept on chip.  next_k1 = (w1, s1, t1, wqkv, bqkv) of the NEXT block: its first half runs on the output
    tile, and its qkv is returned as well."""
    n = p.shape[0]
    out = torch.empty((n, c), dtype=torch.float32, device=p.device)
    arr = (ctypes.c_void_p * 16)(*[(0 if t is None else t.data_ptr()) for t in params])
    tail = [w3, s3, t3, x_res] + (list(next_k1) if next_k1 is not None else [None] * 5)
    tarr = (ctypes.c_void_p * 9)(*[(0 if t is None else t.data_ptr()) for t in tail])
    qn = torch.empty((n, 3 * c), dtype=torch.float32, device=p.device) if next_k1 is not None else None
    base = qkv.data_ptr()
    assert x_res.stride(1) == 1
    _lib.check(_lib.lib().etch_pt_block_k2(n, c, ns, _ptr(p), _vp(base), _vp(base + 4 * c), _vp(base + 8 * c), _c_long(qkv.stride(0)), _ptr(idx), arr,
                                           _ptr(w2), tarr, _c_long(x_res.stride(0)), _ptr(out), _c_long(c), _optptr(qn), _c_long(3 * c), _stream()),
               "etch_pt_block_k2")
    return out, qn


def pt_attention_split(p, qkv, c, idx, params, ns, w2, b2, s3, t3, w5, b5):
    """Same result as pt_attention with the two Linear layers of linear_w on the matrix cores (etch_linear)."""
    n = p.shape[0]
    arr = (ctypes.c_void_p * 16)(*[(0 if t is None else t.data_ptr()) for t in params])
    base = qkv.data_ptr()
    ldq = _c_long(qkv.stride(0))
    w_in = torch.empty((n * ns, c), dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().etch_pt_attn_prep(n, c, ns, _ptr(p), _vp(base), _vp(base + 4 * c), ldq, _ptr(idx), arr, _ptr(w_in), _stream()),
               "etch_pt_attn_prep")
    hid = linear(w_in, w2, bias=b2, scale=s3, shift=t3, act="relu")
    logits = linear(hid, w5, bias=b5)
    out = torch.empty((n, c), dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().etch_pt_attn_aggregate(n, c, ns, _ptr(p), _vp(base + 8 * c), ldq, _ptr(idx), _ptr(logits), arr, _ptr(out), _c_long(c),
                                                 _stream()), "etch_pt_attn_aggregate")
    return out


def pt_group(p, new_p, x, idx, pad_to=1):
    """-> (m*ns, ld) with ld = 3 + c rounded up to a multiple of `pad_to` (padding columns are zero)."""
    m, ns = idx.shape
    c = x.shape[1]
    ld = -(-(3 + c) // pad_to) * pad_to
    out = torch.empty((m * ns, ld), dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().etch_pt_group(m, ns, c, _ptr(p), _ptr(new_p), _ptr(x), _c_long(x.stride(0)), _ptr(idx), _ptr(out), _c_long(ld), _stream()),
               "etch_pt_group")
    return out


def pt_down_gather_max(ux, p, new_p, idx, wp, scale, shift):
    """out[i] = max_j relu(bn(ux[idx[i,j]] + wp @ (p[idx[i,j]] - new_p[i]))): TransitionDown without the grouped rows."""
    m, ns = idx.shape
    co = ux.shape[1]
    out = torch.empty((m, co), dtype=torch.float32, device=ux.device)
    _lib.check(_lib.lib().etch_pt_down_gather_max(m, ns, co, _ptr(ux), _c_long(ux.stride(0)), _ptr(p), _ptr(new_p), _ptr(idx), _ptr(wp), _ptr(scale),
                                                  _ptr(shift), _ptr(out), _stream()), "etch_pt_down_gather_max")
    return out


def gather_rows(x, idx):
    m, c = idx.shape[0], x.shape[1]
    out = torch.empty((m, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_gather_rows(m, c, _ptr(x), _c_long(x.stride(0)), _ptr(idx), _ptr(out), _stream()), "etch_gather_rows")
    return out


def rows_maxpool(x, ns):
    m, c = x.shape[0] // ns, x.shape[1]
    out = torch.empty((m, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_rows_maxpool(m, ns, c, _ptr(x), _ptr(out), _stream()), "etch_rows_maxpool")
    return out


def pt_interp_add(a, f, idx, dist):
    n, c = a.shape
    out = torch.empty_like(a)
    _lib.check(_lib.lib().etch_pt_interp_add(n, c, _ptr(a), _ptr(f), _ptr(idx), _ptr(dist), _ptr(out), _stream()), "etch_pt_interp_add")
    return out


def seg_mean(x, offset, nseg):
    c = x.shape[1]
    out = torch.empty((nseg, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_seg_mean(nseg, c, _ptr(x), _ptr(offset), _ptr(out), _stream()), "etch_seg_mean")
    return out


def concat_bcast(x, g, offset, nseg):
    n, c = x.shape
    out = torch.empty((n, 2 * c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_concat_bcast(n, c, nseg, _ptr(x), _ptr(g), _ptr(offset), _ptr(out), _stream()), "etch_concat_bcast")
    return out


def grouped_dot(h, w, bias, G, J, out=None):
    R = h.shape[0]
    if out is None:
        out = torch.empty((R, G), dtype=torch.float32, device=h.device)
    _lib.check(_lib.lib().etch_grouped_dot(_c_long(R), G, J, _ptr(h), _c_long(h.stride(0)), _ptr(w), _ptr(bias), _ptr(out), _c_long(out.stride(0)),
                                           _stream()), "etch_grouped_dot")
    return out


LRD_SPLIT = os.environ.get("ETCH_LRD_SPLIT", "1") != "0"       # linear_relu_dot on the bf16 matrix cores (split fp32 operands); 0: fp32 MFMA
LRD_F16 = os.environ.get("ETCH_LRD_SPLIT", "1") != "bf16"      # ... as two fp16 planes per operand where the weight-stationary kernel applies; "bf16": three bf16 planes


def lrd_weight_split(w, J=128):
    """[G*J, K] -> int16 [G][K/32][J/16 strips][plane hi / mid / lo][lane = 16 * (k / 8) + column][8]: the exactly split weight of
    etch_linear_relu_dot_split in v_mfma_f32_16x16x32_bf16 B-fragment order."""
    GJ, K = w.shape
    assert GJ % J == 0 and J % 16 == 0 and K % 32 == 0
    planes = split3_bf16(w)                                                   # [3][GJ][K]
    q = planes.reshape(3, GJ // J, J // 16, 16, K // 32, 4, 8)               # [pl][g][strip][col][t][kg][e]
    return q.permute(1, 4, 2, 0, 5, 3, 6).contiguous().reshape(-1)           # [g][t][strip][pl][kg][col][e]


def lrd_weight_split_f16(w, J=128):
    """[G*J, K] -> float16 [G][K/32][J/16 strips][plane h / l][lane = 16 * (k / 8) + column][8]: two fp16 planes of W with every ROW (hidden unit) times
    the power of two that puts its maximum into [8, 16) (exact; the kernel's epilogue takes it out again), in v_mfma_f32_16x16x32_f16 fragment order; the
    rows' inverse powers ride on the tensor as `.wsc` (float32 [G*J], etch_linear_relu_dot_f16)."""
    GJ, K = w.shape
    assert GJ % J == 0 and J % 16 == 0 and K % 32 == 0
    ws, wsc = _row_pow2(w)
    hi = ws.to(torch.float16)
    planes = torch.stack([hi, (ws - hi.float()).to(torch.float16)])          # [2][GJ][K]
    q = planes.reshape(2, GJ // J, J // 16, 16, K // 32, 4, 8)               # [pl][g][strip][col][t][kg][e]
    out = q.permute(1, 4, 2, 0, 5, 3, 6).contiguous().reshape(-1)            # [g][t][strip][pl][kg][col][e]
    out.wsc = wsc
    return out


def permute_weight_frag_grouped(w, J=128):
    """[G*J, K] -> the weight operand of linear_relu_dot: two fp16 planes (lrd_weight_split_f16; round 5) where the weight-stationary kernel covers the
    shape, else the three bf16 planes (lrd_weight_split; int16), else the fp32 fragment order [G][K/16][J/16][64][4] of etch_linear_relu_dot."""
    GJ, K = w.shape
    if LRD_SPLIT and LRD_F16 and J == 128 and K in (32, 64, 128) and (GJ // J == 1 or GJ // J >= 8):
        return lrd_weight_split_f16(w.contiguous(), J)
    if LRD_SPLIT and J == 128 and K in (32, 64, 128, 256):
        return lrd_weight_split(w.contiguous(), J)
    return torch.cat([permute_weight_frag(w[g * J:(g + 1) * J].contiguous()) for g in range(GJ // J)])


def linear_relu_dot(x, w, b1, w2, b2, G, wp=None, out=None):
    """out[r,g] = b2[g] + sum_j relu(x[r] . w[g*J+j] + b1[g*J+j]) * w2[g*J+j]   (J = 128), hidden kept on chip."""
    for t, n in ((w, "w"), (b1, "b1"), (w2, "w2"), (b2, "b2")):
        _need(t, torch.float32, n)
    if x.dtype != torch.float32 or not x.is_cuda:
        raise _lib.EtchHipError("x must be a float32 CUDA(HIP) tensor")
    R, K = x.shape
    J = w.shape[0] // G
    assert x.stride(1) == 1 and w.shape == (G * J, K) and b1.numel() == G * J and w2.numel() == G * J and b2.numel() == G
    if out is None:
        out = torch.empty((R, G), dtype=torch.float32, device=x.device)
    if wp is not None and wp.dtype == torch.float16 and (b1.data_ptr() | w2.data_ptr()) % 16 == 0:
        _lib.check(_lib.lib().etch_linear_relu_dot_f16(_c_long(R), int(K), int(G), int(J), _ptr(x), _c_long(x.stride(0) if R > 1 else K), _ptr(wp), _ptr(wp.wsc),
                                                       _ptr(b1), _ptr(w2), _ptr(b2), _ptr(out), _c_long(out.stride(0) if R > 1 else G), _stream()),
                   "etch_linear_relu_dot_f16")
        return out
    if wp is not None and wp.dtype == torch.float16:
        wp = None                                  # (a bias / w2 view off the 16-byte grid: the fp32 kernel below)
    if wp is not None and wp.dtype == torch.int16:
        _lib.check(_lib.lib().etch_linear_relu_dot_split(_c_long(R), int(K), int(G), int(J), _ptr(x), _c_long(x.stride(0) if R > 1 else K), _ptr(wp),
                                                         _ptr(b1), _ptr(w2), _ptr(b2), _ptr(out), _c_long(out.stride(0) if R > 1 else G), _stream()),
                   "etch_linear_relu_dot_split")
        return out
    _lib.check(_lib.lib().etch_linear_relu_dot(_c_long(R), int(K), int(G), int(J), _ptr(x), _c_long(x.stride(0) if R > 1 else K), _ptr(w),
                                               _c_long(w.stride(0)), _optptr(wp), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(out),
                                               _c_long(out.stride(0) if R > 1 else G), _stream()), "etch_linear_relu_dot")
    return out


def softmax_dot(logits, v):
    R, G = logits.shape
    out = torch.empty((R,), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.lib().etch_softmax_dot(_c_long(R), G, _ptr(logits), _ptr(v), _ptr(out), _stream()), "etch_softmax_dot")
    return out


# ------------------------------------------------------------------ stage 2
def argmax_rows(logits):
    """(..., G) float32 -> (...) int64, first maximum (torch.max semantics)."""
    _need(logits, torch.float32, "logits")
    G = logits.shape[-1]
    R = logits.numel() // G
    out = torch.empty(logits.shape[:-1], dtype=torch.int64, device=logits.device)
    _lib.check(_lib.lib().etch_argmax_rows(_c_long(R), G, _ptr(logits), _ptr(out), _stream()), "etch_argmax_rows")
    return out


def get_markers(points, labels, conf, num_markers):
    _need(points, torch.float32, "points"), _need(labels, torch.int64, "labels"), _need(conf, torch.float32, "conf")
    B, K = labels.shape
    markers = torch.empty((B, num_markers, 3), dtype=torch.float32, device=points.device)
    valid_f = torch.empty((B, num_markers), dtype=torch.float32, device=points.device)
    valid_b = torch.empty((B, num_markers), dtype=torch.bool, device=points.device)
    _lib.check(_lib.lib().etch_get_markers(B, K, num_markers, _ptr(points), _ptr(labels), _ptr(conf), _ptr(markers), _ptr(valid_f),
                                           _ptr(valid_b), _stream()), "etch_get_markers")
    return markers, valid_f, valid_b


LM_SPLIT_MAX_BATCH = 8     # scans per launch up to which a scan's linearisation is split over several workgroups (latency regime)
LM_SPLIT_WGS = 3           # SMPL: 3 marker chunks of 32 -> 3 workgroups (measured: 6.4 -> 5.2 ms per 30+50 fit; 4+ only add exchange cost)
LM_SPLIT_WGS_BY_JOINTS = {55: 4}      # SMPL-X-sized: 10 chunks of 9 markers; 58 -> 45.6 ms per 75+125 fit at B = 8 (G = 5: 42.5 ms at B = 1)


def lm_split_default(B, nj):
    return LM_SPLIT_WGS_BY_JOINTS.get(int(nj), LM_SPLIT_WGS) if B <= LM_SPLIT_MAX_BATCH else 1


def smpl_lm_fit(consts, markers, valid_f, it0, step0, damp0, it1, step1, damp1, want_trace=False, phase_ticks=None, nj=24, nb=10, split=None):
    """x (B, 3 nj + nb + 3) = pose | betas | orient | transl.  (nj, nb) = (24, 10) SMPL or (55, 20) SMPL-X-sized.
    split = workgroups per scan (None: lm_split_default -- 3 / 4 when B <= LM_SPLIT_MAX_BATCH, else one persistent workgroup per scan)."""
    B, M = valid_f.shape
    dof = 3 * nj + nb + 3
    arr = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in consts])
    x = torch.empty((B, dof), dtype=torch.float32, device=markers.device)
    x0 = torch.empty((B, dof), dtype=torch.float32, device=markers.device)
    tr = torch.zeros((B, it0 + it1 + 2), dtype=torch.float32, device=markers.device) if want_trace else None
    G = int(split) if split is not None else lm_split_default(B, nj)
    if G > 1 and B * G > torch.cuda.get_device_properties(markers.device).multi_processor_count:
        G = 1           # the workgroups of a scan wait for each other: on a small part / CU partition one persistent workgroup per scan (same results up to summation order)
    ws = None
    if G > 1:
        nbytes = _lib.lib().etch_smpl_lm_split_workspace_bytes(B, int(nj), int(nb), G)
        ws = torch.empty((nbytes // 8 + 1,), dtype=torch.float64, device=markers.device)
    _lib.check(_lib.lib().etch_smpl_lm_fit_split(B, M, int(nj), int(nb), arr, _ptr(markers), _ptr(valid_f), int(it0), _c_float(step0), _c_float(damp0),
                                                 int(it1), _c_float(step1), _c_float(damp1), _ptr(x), _ptr(x0), _optptr(tr), _optptr(phase_ticks),
                                                 G, _optptr(ws), _stream()), "etch_smpl_lm_fit_split")
    x._etch_split_ws = ws       # the split fit's workspace carries the per-scan give-up flags (smpl_lm_split_failed); None for one workgroup per scan
    return x, x0, tr


def smpl_lm_split_failed(x):
    """Number of scans of a split LM fit that were abandoned because a partner workgroup never arrived (their rows of `x` are NaN).  `x` = the
    first tensor smpl_lm_fit returned.  Synchronises the current stream."""
    ws = getattr(x, "_etch_split_ws", None)
    if ws is None:
        return 0
    n = ctypes.c_int(0)
    _lib.check(_lib.lib().etch_smpl_lm_split_failed(int(x.shape[0]), _ptr(ws), ctypes.byref(n), _stream()), "etch_smpl_lm_split_failed")
    return int(n.value)


def smpl_adam_fit(consts, markers, valid_f, it0, it1, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, want_trace=False, nj=24, nb=10):
    """First-order fitter (fit_SMPL_Adam.py): -> x (B,DOF), x_last (parameters of the last forward pass), loss trace (B, it0+it1) or None."""
    B, M = valid_f.shape
    dof = 3 * nj + nb + 3
    arr = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in consts])
    x = torch.empty((B, dof), dtype=torch.float32, device=markers.device)
    xl = torch.empty((B, dof), dtype=torch.float32, device=markers.device)
    tr = torch.zeros((B, it0 + it1), dtype=torch.float32, device=markers.device) if want_trace else None
    _lib.check(_lib.lib().etch_smpl_adam_fit(B, M, int(nj), int(nb), arr, _ptr(markers), _ptr(valid_f), int(it0), int(it1), _c_float(lr),
                                             _c_float(beta1), _c_float(beta2), _c_float(eps), _ptr(x), _ptr(xl), _optptr(tr), _stream()),
               "etch_smpl_adam_fit")
    return x, xl, tr


def marker_status(markers, valid_f):
    """(B,) int32: bit 0 = a valid marker is non-finite (the fit of that scan is NaN, as in the reference), bit 1 = no valid marker."""
    B, M = valid_f.shape
    st = torch.empty((B,), dtype=torch.int32, device=markers.device)
    _lib.check(_lib.lib().etch_marker_status(B, M, _ptr(markers), _ptr(valid_f), _ptr(st), _stream()), "etch_marker_status")
    return st


def smpl_lm_linearize(consts, x, markers, valid_f, nb_active, nj=24, nb=10, want_normal=False):
    """Diagnostics: residual (B,3M), analytic Jacobian (B,3M,DOF) and optionally the matrix-core normal equations (B,DOF+1,DOF+1) fp64."""
    B, M = valid_f.shape
    dof = 3 * nj + nb + 3
    assert x.shape == (B, dof)
    arr = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in consts])
    r = torch.empty((B, 3 * M), dtype=torch.float32, device=x.device)
    J = torch.empty((B, 3 * M, dof), dtype=torch.float32, device=x.device)
    N = torch.empty((B, dof + 1, dof + 1), dtype=torch.float64, device=x.device) if want_normal else None
    _lib.check(_lib.lib().etch_smpl_lm_linearize(B, M, int(nj), int(nb), int(nb_active), arr, _ptr(x), _ptr(markers), _ptr(valid_f), _ptr(r),
                                                 _ptr(J), _optptr(N), _stream()), "etch_smpl_lm_linearize")
    return (r, J, N) if want_normal else (r, J)


def rodrigues(theta):
    """theta (n,3) fp32 -> R (n,3,3) fp64, dR (n,3,3,3) with dR[n,q] = dR/dtheta_q."""
    _need(theta, torch.float32, "theta")
    n = theta.shape[0]
    R = torch.empty((n, 3, 3), dtype=torch.float64, device=theta.device)
    dR = torch.empty((n, 3, 3, 3), dtype=torch.float32, device=theta.device)
    _lib.check(_lib.lib().etch_rodrigues(n, _ptr(theta), _ptr(R), _ptr(dR), _stream()), "etch_rodrigues")
    return R, dR


def smpl_lbs(consts, x, V, n_extra, nj=24, nb=10):
    B = x.shape[0]
    assert x.shape[1] == 3 * nj + nb + 3
    arr = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in consts])
    verts = torch.empty((B, V, 3), dtype=torch.float32, device=x.device)
    joints = torch.empty((B, nj + n_extra, 3), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().etch_smpl_lbs(B, V, int(nj), int(nb), n_extra, arr, _ptr(x), _ptr(verts), _ptr(joints), _stream()), "etch_smpl_lbs")
    return verts, joints


def inner_points(points, direction, magnitude, scale):
    _need(points, torch.float32, "points"), _need(direction, torch.float32, "direction"), _need(magnitude, torch.float32, "magnitude")
    out = torch.empty_like(points)
    n = points.numel() // 3
    _lib.check(_lib.lib().etch_inner_points(_c_long(n), _ptr(points), _ptr(direction), _ptr(magnitude), _c_float(scale), _ptr(out), _stream()),
               "etch_inner_points")
    return out
