"""Differentiable forms of the encoder's native ops: torch.autograd.Function wrappers whose forward is the fused
inference kernel and whose backward is hand-written HIP (csrc/backward.hip) -- what /root/reference/src/train.py:77-101 obtains from
autograd through the un-fused vgtk forms (external/vgtk/vgtk/so3conv/functional.py:224-378, modules.py:33-39).

Gradients are reproducible run to run (fixed reduction orders, no atomics).  Memory: the backward of the inter conv regenerates the
kernel weights and the grouped features for `chunk` output points at a time instead of keeping the reference's [b,p,60,24,nn] and
[b,c,24,p,60] tensors alive between forward and backward."""
import ctypes

import torch

from . import _lib
from . import ops
from .ops import _c_float, _ptr, _stream


def _check(status, what):
    _lib.check(status, what)


import os

SLOT_DFEAT = os.environ.get("ETCH_SLOT_DFEAT", "1") != "0"      # the inter conv's feature gradient from the target side (round 6; 0: inter_dfeat_kernel)
CHUNK_BYTES = 1.2e9                                             # bound on each of a chunk's temporaries in the inter conv's backward
_COUNTERS = {}


def reduce_counters(device):
    """The 64 zero-initialised counters of the one-launch reductions (include/etch_hip.h: etch_gemm_tn_fused, etch_colsum_fused, etch_bn_*): every call
    leaves them zero; one set per (device, stream) -- calls that share a set must be ordered on one stream, which launches on one stream are."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    c = _COUNTERS.get(key)
    if c is None:
        c = _COUNTERS[key] = torch.zeros((64,), dtype=torch.int32, device=device)
    return c


def gemm_tn(A, B, out=None, accumulate=False):
    """A (R,M), B (R,N) -> A^T B (M,N)."""
    R, M = A.shape
    N = B.shape[1]
    C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    ws = torch.empty((_lib.lib().etch_gemm_tn_workspace_floats(ctypes.c_long(R), M, N),), dtype=torch.float32, device=A.device)
    counters = reduce_counters(A.device)
    _check(_lib.lib().etch_gemm_tn_fused(ctypes.c_long(R), M, N, _ptr(A), ctypes.c_long(A.stride(0)), _ptr(B), ctypes.c_long(B.stride(0)), _ptr(C),
                                         1 if accumulate else 0, _ptr(ws), _ptr(counters), _stream()), "etch_gemm_tn_fused")
    return C


def colsum(x2d):
    R, C = x2d.shape
    ws = torch.empty((64 * C,), dtype=torch.float64, device=x2d.device)
    out = torch.empty((C,), dtype=torch.float32, device=x2d.device)
    counters = reduce_counters(x2d.device)
    _check(_lib.lib().etch_colsum_fused(ctypes.c_long(R), C, _ptr(x2d), ctypes.c_long(x2d.stride(0)), _ptr(ws), _ptr(counters), _ptr(out),
                                        _stream()), "etch_colsum_fused")
    return out


class InterSO3ConvFunction(torch.autograd.Function):
    """y (b,p2,60,cout) = fused inter conv of feats (b,p1,60,cin) [channels-last] with W (cout, cin*24), bias (cout)."""

    @staticmethod
    def forward(ctx, feats_cl, W, bias, xyz, new_xyz, ball_idx, rk, sigma, chunk):
        feats_cl, Wc, bc = feats_cl.contiguous(), W.detach().contiguous(), bias.detach().reshape(-1).contiguous()
        cin = feats_cl.shape[-1]
        Wp = ops.inter_weight_frag(Wc, cin) if cin % 16 == 0 else None
        y = ops.inter_so3conv(xyz, new_xyz, ball_idx, feats_cl, rk, Wc, Wp, bc, sigma)
        ctx.save_for_backward(feats_cl, Wc, xyz, new_xyz, ball_idx, rk)
        ctx.sigma, ctx.chunk, ctx.bias_shape = float(sigma), int(chunk), bias.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        feats, W, xyz, new_xyz, idx, rk = ctx.saved_tensors
        dy = dy.contiguous()
        b, p1, na, cin = feats.shape
        p2, nn = idx.shape[1], idx.shape[2]
        cout, kk = W.shape
        lib = _lib.lib()
        need_df = ctx.needs_input_grad[0]
        dW = torch.zeros_like(W)
        dfeats = None
        Wt = W.t().contiguous()                                              # (kk, cout): dX1 = dY W  as an NT product
        # chunks of output points: the X1 / dX1 rows (b pc 60 x cin 24) and the per-slot contributions (b pc nn x 60 cin) of a chunk stay under ~1.2 GB each;
        # one chunk where that allows (a training batch of one scan: one sort of the slots per conv and step)
        slots = SLOT_DFEAT and nn <= 64
        per_point = 4 * b * max(na * kk, nn * na * cin if slots else 0)
        chunk = max(ctx.chunk, min(p2, int(CHUNK_BYTES // per_point)))
        for s in range(0, p2, chunk):
            pc = min(chunk, p2 - s)
            x1 = torch.empty((b, pc, na, kk), dtype=torch.float32, device=feats.device)
            _check(lib.etch_inter_x1_rows(b, cin, p1, p2, s, pc, nn, _c_float(ctx.sigma), _ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(feats), _ptr(rk),
                                          _ptr(x1), _stream()), "etch_inter_x1_rows")
            dyc = dy[:, s:s + pc].contiguous().view(b * pc * na, cout)
            gemm_tn(dyc, x1.view(b * pc * na, kk), out=dW, accumulate=True)   # dW += dY^T X1
            if need_df and slots:
                dx1 = ops.linear(dyc, Wt)                                    # (rows, kk) = dY W
                contrib = torch.empty((b * pc * nn, na * cin), dtype=torch.float32, device=feats.device)
                _check(lib.etch_inter_dfeat_slots(b, cin, p1, p2, s, pc, nn, _c_float(ctx.sigma), _ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(rk), _ptr(dx1),
                                                  _ptr(contrib), _stream()), "etch_inter_dfeat_slots")
                del dx1
                # rows of one source point, in slot order (stable sort): the reproducible scatter-add; padded slots (index < 0) go to a segment of their own
                src = idx[:, s:s + pc].long() + (torch.arange(b, device=idx.device, dtype=torch.int64) * p1).view(b, 1, 1)
                src = torch.where(idx[:, s:s + pc] < 0, torch.full_like(src, b * p1), src).reshape(-1)
                part = segment_sum_rows(contrib, src, b * p1 + 1)[:b * p1].view(b, p1, na, cin)
                dfeats = part if dfeats is None else dfeats.add_(part)
                del contrib
            elif need_df:
                if dfeats is None:
                    dfeats = torch.zeros_like(feats)
                dx1 = ops.linear(dyc, Wt)                                    # (rows, kk) = dY W
                _check(lib.etch_inter_dfeat(b, cin, p1, p2, s, pc, nn, _c_float(ctx.sigma), _ptr(xyz), _ptr(new_xyz), _ptr(idx), _ptr(rk), _ptr(dx1),
                                            _ptr(dfeats), 1, _stream()), "etch_inter_dfeat")
        dbias = colsum(dy.view(-1, cout)).view(ctx.bias_shape)
        return dfeats, dW, dbias, None, None, None, None, None, None


def inter_so3conv(feats_cl, W, bias, xyz, new_xyz, ball_idx, rk, sigma, chunk=1024):
    return InterSO3ConvFunction.apply(feats_cl, W, bias, xyz, new_xyz, ball_idx, rk, sigma, chunk)


_INV_TABLES = {}


def _inverse_tables(idx32):
    """Column-wise inverse of the intra conv's anchor tables (60, nt) -- every column is a permutation of the anchors.  A constant of the module's buffer:
    built once per (tensor, version) instead of with 2 nt launches in every backward (96 per training step)."""
    key = (idx32.data_ptr(), idx32._version, tuple(idx32.shape), str(idx32.device))
    hit = _INV_TABLES.get(key)
    if hit is None:
        if len(_INV_TABLES) > 16:
            _INV_TABLES.clear()
        with torch.no_grad():
            inv = torch.empty_like(idx32)
            ar = torch.arange(idx32.shape[0], dtype=torch.int32, device=idx32.device)
            for t in range(idx32.shape[1]):
                inv[idx32[:, t].long(), t] = ar
        hit = _INV_TABLES[key] = (inv.contiguous(), idx32)          # (the key tensor is held: its address stays unique)
    return hit[0]


class IntraSO3ConvFunction(torch.autograd.Function):
    """y (b,p,60,C) = fused intra conv of x (b,p,60,C) with W (C, C*12) [column c*12+t], bias (C); intra_idx (60,12) whose columns are
    permutations of the anchors (so the data gradient is the same kernel run with the inverse tables and the transposed weight)."""

    @staticmethod
    def _frag(W, C, nt):
        W2 = W.view(W.shape[0], C, nt).permute(0, 2, 1).reshape(W.shape[0], nt * C).contiguous()     # tap-major K, as the kernel reads it
        return ops.permute_weight_frag(W2)

    @staticmethod
    def forward(ctx, x_cl, W, bias, intra_idx):
        x_cl, Wc = x_cl.contiguous(), W.detach().contiguous()
        C, nt = x_cl.shape[-1], intra_idx.shape[1]
        idx32 = intra_idx.to(torch.int32).contiguous()
        y = ops.intra_so3conv(x_cl, idx32, IntraSO3ConvFunction._frag(Wc, C, nt), bias.detach().reshape(-1).contiguous(), Wc.shape[0])
        ctx.save_for_backward(x_cl, Wc, idx32)
        ctx.bias_shape = bias.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, idx32 = ctx.saved_tensors
        dy = dy.contiguous()
        b, p, na, C = x.shape
        nt = idx32.shape[1]
        cout = W.shape[0]
        lib = _lib.lib()
        dx = None
        if ctx.needs_input_grad[0]:
            # dx[p,a',c] = sum_t sum_o W[o, c*nt+t] dy[p, inv_t(a'), o]: the forward kernel with inverse tables and W'[c, o*nt+t] = W[o, c*nt+t]
            inv = _inverse_tables(idx32)
            Wb = W.view(cout, C, nt).permute(1, 0, 2).reshape(C, cout * nt).contiguous()
            zero = torch.zeros((C,), dtype=torch.float32, device=x.device)
            dx = ops.intra_so3conv(dy, inv, IntraSO3ConvFunction._frag(Wb, cout, nt), zero, C)
        xg = torch.empty((b * p * na, nt * C), dtype=torch.float32, device=x.device)
        _check(lib.etch_intra_rows(ctypes.c_long(b * p), C, nt, _ptr(idx32), _ptr(x), _ptr(xg), _stream()), "etch_intra_rows")
        dW2 = gemm_tn(dy.view(b * p * na, cout), xg)                          # (cout, nt*C), tap-major columns
        dW = dW2.view(cout, nt, C).permute(0, 2, 1).reshape(cout, C * nt).contiguous()
        dbias = colsum(dy.view(-1, cout)).view(ctx.bias_shape)
        return dx, dW, dbias, None


def intra_so3conv(x_cl, W, bias, intra_idx):
    return IntraSO3ConvFunction.apply(x_cl, W, bias, intra_idx)


class InstanceNormLeakyReLUFunction(torch.autograd.Function):
    """leaky_relu(InstanceNorm2d(affine=False, eps=1e-5)(x), 0.01) on channels-last x (b, ..., C) (so3conv.py:36-44)."""

    @staticmethod
    def forward(ctx, x_cl, slope):
        x_cl = x_cl.contiguous()
        mean, rstd = ops.instnorm_stats(x_cl)
        assert abs(slope - 0.01) < 1e-12, "the fused forward kernel applies leaky_relu(0.01)"
        y = ops.instnorm_act_add(x_cl, mean, rstd)
        ctx.save_for_backward(x_cl, mean, rstd)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        b, C = x.shape[0], x.shape[-1]
        rows = x.numel() // (b * C)
        ws = torch.empty((_lib.lib().etch_instnorm_act_backward_workspace_bytes(b, C) // 8,), dtype=torch.float64, device=x.device)
        dx = torch.empty_like(x)
        _check(_lib.lib().etch_instnorm_act_backward(b, rows, C, _ptr(x), _ptr(dy), _ptr(mean), _ptr(rstd), _c_float(ctx.slope), _ptr(ws), _ptr(dx),
                                                     _stream()), "etch_instnorm_act_backward")
        return dx, None


def instnorm_leaky_relu(x_cl, slope=0.01):
    return InstanceNormLeakyReLUFunction.apply(x_cl, slope)


def _mhsa_forward(x2, wq, wk, wv, wc, bc, mode):
    """One MultiHeadAttention layer on tokens x2 (T*60, E): the fused inference kernel for E = 64 (etch_mhsa_layer), the un-fused chain
    Linear -> etch_mhsa_attention_dim -> Linear for the other encoder depths (E = 32 / 128 / 256: models_pointcloud.py:34-48).
    mode 0 residual, 1 plain, 2 concatenated heads only."""
    E = x2.shape[1]
    T = x2.shape[0] // 60
    if E == 64:
        return ops.mhsa_layer(x2, wq, wk, wv, wc, bc, mode=mode)
    qkv = ops.linear(x2, torch.cat([wq, wk, wv], 0).contiguous())
    att = ops.mhsa_attention(qkv, T, 0, E, 2 * E, embedding_dim=E)
    if mode == 2:
        return att
    return ops.linear(att, wc, bias=bc, res=x2 if mode == 0 else None, res_mode=2 if mode == 0 else 0)


def _mhsa_attention_backward(T, E, qkv, dO):
    dqkv = torch.empty_like(qkv)
    _check(_lib.lib().etch_mhsa_attention_backward_dim(ctypes.c_long(T), int(E), _ptr(qkv), ctypes.c_long(3 * E), 0, E, 2 * E, _ptr(dO), ctypes.c_long(E),
                                                       _ptr(dqkv), _stream()), "etch_mhsa_attention_backward")
    return dqkv


class MHSALayerFunction(torch.autograd.Function):
    """One MultiHeadAttention layer of the direction head (direction_backbones.py:132-194, + the residual of :216-221): forward = the
    fused inference kernel (etch_mhsa_layer; the un-fused chain for token widths other than 64); backward recomputes q|k|v and the head
    outputs with the un-fused kernels, runs the attention core's hand-written backward (etch_mhsa_attention_backward_dim) and closes the
    linear maps with the matrix-core GEMMs."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, wc, bc, residual):
        T, E = x.shape[0], x.shape[-1]
        x2 = x.reshape(T * 60, E).contiguous()
        ws = [t.detach().contiguous() for t in (wq, wk, wv, wc)]
        y = _mhsa_forward(x2, ws[0], ws[1], ws[2], ws[3], bc.detach().contiguous(), 0 if residual else 1)
        ctx.save_for_backward(x2, *ws)
        ctx.residual, ctx.xshape = bool(residual), x.shape
        return y.view(T, 60, wc.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wq, wk, wv, wc = ctx.saved_tensors
        T, E = x2.shape[0] // 60, x2.shape[1]
        dy2 = dy.reshape(T * 60, wc.shape[0]).contiguous()
        wqkv = torch.cat([wq, wk, wv], 0).contiguous()                       # (3E, E)
        qkv = ops.linear(x2, wqkv)                                           # recomputed, not kept from the forward
        att = ops.mhsa_attention(qkv, T, 0, E, 2 * E, embedding_dim=E)
        dO = ops.linear(dy2, wc.t().contiguous())                            # dY Wc
        dqkv = _mhsa_attention_backward(T, E, qkv, dO)
        dx = ops.linear(dqkv, wqkv.t().contiguous(), res=dy2 if ctx.residual else None, res_mode=2 if ctx.residual else 0)
        dwqkv = gemm_tn(dqkv, x2)                                            # (3E, E)
        dwc = gemm_tn(dy2, att)
        return dx.view(ctx.xshape), dwqkv[:E], dwqkv[E:2 * E], dwqkv[2 * E:], dwc, colsum(dy2), None


def mhsa_layer(x, wq, wk, wv, wc, bc, residual=True):
    """x (T,60,E) -> (T,60,E), differentiable in x and all five parameters; E in {32, 64, 128, 256}."""
    return MHSALayerFunction.apply(x, wq, wk, wv, wc, bc, residual)


def segment_sum_rows(src, index, nseg):
    """dst[q] = sum of the rows e of src with index[e] == q, summed in increasing e: scatter-add with a fixed order."""
    key, perm = torch.sort(index.reshape(-1).long(), stable=True)
    seg = torch.searchsorted(key, torch.arange(nseg + 1, device=key.device, dtype=torch.int64)).contiguous()
    dst = torch.empty((nseg, src.shape[1]), dtype=torch.float32, device=src.device)
    _check(_lib.lib().etch_segment_sum_rows(ctypes.c_long(nseg), src.shape[1], _ptr(src), _ptr(perm.contiguous()), _ptr(seg), _ptr(dst), _stream()),
           "etch_segment_sum_rows")
    return dst


class PTVectorAttentionFunction(torch.autograd.Function):
    """The vector-attention core of PointTransformerLayer (pointtransformer_seg.py:28-36 after the q / k / v Linear layers) with eval-mode
    BatchNorm as folded per-channel (scale, shift) constants.  Differentiable in xq, xk, xv and in the Linear layers of linear_p / linear_w;
    forward = etch_pt_attention, backward = etch_pt_attention_backward + matrix-core GEMMs / column sums / ordered segment sums."""

    @staticmethod
    def forward(ctx, p, xq, xk, xv, idx, W0, b0, s_p, t_p, W3, b3, s_w0, t_w0, W2, b2, s_w3, t_w3, W5, b5):
        d = lambda t: t.detach().contiguous()
        n, c = xq.shape
        ns = idx.shape[1]
        qkv = torch.cat([d(xq), d(xk), d(xv)], 1).contiguous()
        params = [d(W0), d(b0), d(s_p), d(t_p), d(W3), d(b3), d(s_w0), d(t_w0), d(W2).t().contiguous(), d(b2), d(s_w3), d(t_w3), d(W5), d(b5)]
        out = ops.pt_attention(p, qkv, c, idx, params + [None, None], ns)
        ctx.save_for_backward(p, qkv, idx, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, qkv, idx, *params = ctx.saved_tensors
        n, ns = idx.shape
        c = qkv.shape[1] // 3
        cs, E = c // 8, n * ns
        dev = qkv.device
        dout = dout.contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        GV, GU, A, DZ, G, DL, H4, DH4, R4, dxq = (new(E, c), new(E, c), new(E, c), new(E, cs), new(E, cs), new(E, cs), new(E, 4), new(E, 4),
                                                  new(E, 4), new(n, c))
        outs = [GV, GU, A, DZ, G, DL, H4, DH4, R4, dxq]
        parr = (ctypes.c_void_p * 16)(*([t.data_ptr() for t in params] + [0, 0]))
        oarr = (ctypes.c_void_p * 10)(*[t.data_ptr() for t in outs])
        base = qkv.data_ptr()
        vp = ctypes.c_void_p
        _check(_lib.lib().etch_pt_attention_backward(n, c, ns, _ptr(p), vp(base), vp(base + 4 * c), vp(base + 8 * c), ctypes.c_long(qkv.stride(0)),
                                                     _ptr(idx), parr, _ptr(dout), ctypes.c_long(dout.stride(0)), oarr, _stream()),
               "etch_pt_attention_backward")
        dxk = segment_sum_rows(GU, idx, n)
        dxv = segment_sum_rows(GV, idx, n)
        dW5, db5 = gemm_tn(DL, G), colsum(DL)
        dW2, db2 = gemm_tn(DZ, A), colsum(DZ)
        dW3 = gemm_tn(GU, H4, out=gemm_tn(GV, H4), accumulate=True)[:, :3].contiguous()
        db3 = colsum(GV) + colsum(GU)
        dW0, db0 = gemm_tn(DH4, R4)[:3, :3].contiguous(), colsum(DH4)[:3].contiguous()
        return (None, dxq, dxk, dxv, None, dW0, db0, None, None, dW3, db3, None, None, dW2, db2, None, None, dW5, db5)


def pt_vector_attention(p, xq, xk, xv, idx, W0, b0, s_p, t_p, W3, b3, s_w0, t_w0, W2, b2, s_w3, t_w3, W5, b5):
    """p (n,3), xq / xk / xv (n,c), idx (n,ns) int32 kNN indices, linear_p = [W0 (3,3), b0, BN(s_p, t_p), ReLU, W3 (c,3), b3],
    linear_w = [BN(s_w0, t_w0), ReLU, W2 (c/8,c), b2, BN(s_w3, t_w3), ReLU, W5 (c/8,c/8), b5] -> (n,c)."""
    return PTVectorAttentionFunction.apply(p, xq, xk, xv, idx, W0, b0, s_p, t_p, W3, b3, s_w0, t_w0, W2, b2, s_w3, t_w3, W5, b5)


# ------------------------------------------------------------------------------------------------ the rest of the encoder + direction head
class LinearFunction(torch.autograd.Function):
    """y = act(x W^T + b) on the fp32 matrix cores (etch_linear), differentiable in x, W, b.  act in (None, "relu")."""

    @staticmethod
    def forward(ctx, x2, W, bias, act):
        x2, Wc = x2.contiguous(), W.detach().contiguous()
        y = ops.linear(x2, Wc, bias=None if bias is None else bias.detach().contiguous(), act=act)
        ctx.save_for_backward(x2, Wc, y if act == "relu" else None)
        ctx.act, ctx.has_bias = act, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.act == "relu":
            dy = dy * (y > 0)
        dx = ops.linear(dy, W.t().contiguous()) if ctx.needs_input_grad[0] else None
        if x2.shape[1] == 1:                     # one input channel (the first skip conv): dW[o] = sum_r dy[r,o] x[r]
            dW = colsum(dy * x2).view(-1, 1)
        else:
            dW = gemm_tn(dy, x2)
        db = colsum(dy) if ctx.has_bias else None
        return dx, dW, db, None


def linear(x, W, bias=None, act=None):
    """x (..., K) -> (..., O)."""
    shp = x.shape
    y = LinearFunction.apply(x.reshape(-1, shp[-1]), W, bias, act)
    return y.view(*shp[:-1], W.shape[0])


class GatherPointsFunction(torch.autograd.Function):
    """x (b, p1, ...) -> x[b, idx[b, :]] (b, p2, ...) for per-sample index lists WITHOUT repeats (furthest-point samples, lazy prefixes):
    the skip branch's sub-sampling (so3conv.py:178-180)."""

    @staticmethod
    def forward(ctx, x, idx):
        b = x.shape[0]
        ar = torch.arange(b, device=x.device).view(b, 1)
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return x[ar, idx.long()].contiguous()

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        b = ctx.shape[0]
        dx = torch.zeros(ctx.shape, dtype=dy.dtype, device=dy.device)
        dx[torch.arange(b, device=dy.device).view(b, 1), idx.long()] = dy        # unique indices: a plain (deterministic) scatter
        return dx, None


class PropInterpFunction(torch.autograd.Function):
    """3-NN feature propagation on channels-last features (pointnet2_utils.py:45-74): feats_cl (B,S,A,C), idx / w (B,N,3) -> (B,N,A,C).
    No gradient into idx / w (functions of the coordinates)."""

    @staticmethod
    def forward(ctx, feats_cl, idx, w):
        feats_cl = feats_cl.contiguous()
        out, _ = ops.prop_interp(feats_cl, idx, w)
        ctx.save_for_backward(idx, w)
        ctx.shape = feats_cl.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, w = ctx.saved_tensors
        B, S, A, C = ctx.shape
        N = idx.shape[1]
        dout = dout.contiguous()
        key = (idx.long() + (torch.arange(B, device=idx.device) * S).view(B, 1, 1)).reshape(-1)
        skey, perm = torch.sort(key, stable=True)
        seg = torch.searchsorted(skey, torch.arange(B * S + 1, device=key.device, dtype=torch.int64)).contiguous()
        dfe = torch.empty((B * S, A * C), dtype=torch.float32, device=dout.device)
        _check(_lib.lib().etch_weighted_segment_sum_rows(ctypes.c_long(B * S), A * C, 3, _ptr(dout), _ptr(w.contiguous()), _ptr(perm.contiguous()), _ptr(seg),
                                                         _ptr(dfe), _stream()), "etch_weighted_segment_sum_rows")
        return dfe.view(B, S, A, C), None, None


def prop_interp(feats_cl, idx, w):
    return PropInterpFunction.apply(feats_cl, idx, w)


class MHSAHeadsFunction(torch.autograd.Function):
    """The concatenated head outputs of one MultiHeadAttention layer (before head_combine): x (T,60,E) -> (T,60,E); forward = the fused
    kernel in mode 2 (E = 64) or the un-fused chain, backward as MHSALayerFunction without the combine."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv):
        T, E = x.shape[0], x.shape[-1]
        x2 = x.reshape(T * 60, E).contiguous()
        ws = [t.detach().contiguous() for t in (wq, wk, wv)]
        y = _mhsa_forward(x2, ws[0], ws[1], ws[2], None, None, 2)
        ctx.save_for_backward(x2, *ws)
        ctx.xshape = x.shape
        return y.view(T, 60, E)

    @staticmethod
    def backward(ctx, dO):
        x2, wq, wk, wv = ctx.saved_tensors
        T, E = x2.shape[0] // 60, x2.shape[1]
        dO2 = dO.reshape(T * 60, E).contiguous()
        wqkv = torch.cat([wq, wk, wv], 0).contiguous()
        qkv = ops.linear(x2, wqkv)
        dqkv = _mhsa_attention_backward(T, E, qkv, dO2)
        dx = ops.linear(dqkv, wqkv.t().contiguous())
        dwqkv = gemm_tn(dqkv, x2)
        return dx.view(ctx.xshape), dwqkv[:E], dwqkv[E:2 * E], dwqkv[2 * E:]


def mhsa_heads(x, wq, wk, wv):
    return MHSAHeadsFunction.apply(x, wq, wk, wv)


class SO3MeanDirFunction(torch.autograd.Function):
    """w (T,60) -> R(sum_a w_a R_a) @ [0,0,1] (T,3): so3_mean (so3conv.py:186-225) followed by the rotation of the standard vector
    (models_pointcloud.py:120-124); backward = etch_so3_mean_dir_backward (derivative of the polar factor)."""

    @staticmethod
    def forward(ctx, w, anchors):
        w, anchors = w.contiguous(), anchors.contiguous()
        d, _, _ = ops.so3_mean_dir(w, anchors)
        ctx.save_for_backward(w, anchors)
        return d

    @staticmethod
    def backward(ctx, dd):
        w, anchors = ctx.saved_tensors
        T, A = w.shape
        dw = torch.empty_like(w)
        _check(_lib.lib().etch_so3_mean_dir_backward(ctypes.c_long(T), A, _ptr(w), _ptr(anchors), _ptr(dd.contiguous()), _ptr(dw), _stream()),
               "etch_so3_mean_dir_backward")
        return dw, None


def so3_mean_dir(w, anchors):
    return SO3MeanDirFunction.apply(w, anchors)
