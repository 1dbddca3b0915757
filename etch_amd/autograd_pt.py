"""Differentiable forms of the two Point-Transformer nets: what /root/reference/src/train.py:77-101 back-propagates through
/root/reference/src/models/pointtransformer_seg.py -- in train() mode (BatchNorm1d on batch statistics, running statistics updated) or in
eval() mode (running statistics as constants).

The nets are evaluated UN-fused here: the inference kernels fold eval-mode BatchNorm into per-channel constants inside the fused
attention / TransitionDown kernels, which has no meaning when the statistics are functions of the batch.  Every dense layer, gather,
segment sum, BatchNorm reduction, max-pool routing and the softmax-aggregation run on kernels of libetch_hip.so (etch_linear / etch_gemm_tn,
etch_gather_rows / etch_segment_sum_rows, csrc/train_ops.hip); broadcasts, adds and the 3 x 3 position encoder's first layer are torch
element-wise ops.  FPS / kNN indices are functions of the coordinates only and carry no gradient (as in the reference, whose index
kernels are not differentiable either).  Gradients are reproducible run to run (fixed-order reductions, no atomics)."""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from . import autograd as A
from . import ops
from .models import pointops
from .ops import _ptr, _stream

_c_long = ctypes.c_long


def _ws(C, device):
    return torch.empty((64 * 2 * C,), dtype=torch.float64, device=device)


def bn_stats(x2):
    """Column mean and biased variance of x2 (R,C), fp64 sums in a fixed order."""
    R, C = x2.shape
    mean = torch.empty((C,), dtype=torch.float32, device=x2.device)
    var = torch.empty_like(mean)
    ws = _ws(C, x2.device)               # held in a name: a temporary would be back in the allocator (and could be handed out again) before the launch
    _lib.check(_lib.lib().etch_bn_stats(_c_long(R), C, _ptr(x2), _c_long(x2.stride(0)), _ptr(ws), _ptr(mean), _ptr(var), _stream()),
               "etch_bn_stats")
    return mean, var


class BatchNormFunction(torch.autograd.Function):
    """y = act(gamma * (x - mean) / sqrt(var + eps) + beta) on rows x (R,C).  train: (mean, var) are the batch statistics of x (the backward
    carries their dependence on x), computed here together with the module's running statistics (etch_bn_train_forward: 2 launches; round 5 spent
    12 per call -- 246 calls per training step); eval: (mean, var) given, constants."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, relu, train, running):
        x = x.contiguous()
        R, C = x.shape
        g = gamma.detach().contiguous()
        y = torch.empty((R, C), dtype=torch.float32, device=x.device)
        if train:
            rm, rv, nbt, momentum = running
            stats = torch.empty((3, C), dtype=torch.float32, device=x.device)
            ws, counters = _ws(C, x.device), A.reduce_counters(x.device)
            mean, rstd, scale = stats[0], stats[1], stats[2]
            _lib.check(_lib.lib().etch_bn_train_forward(_c_long(R), C, _ptr(x), _c_long(x.stride(0)), _ptr(g), _ptr(beta.detach().contiguous()),
                                                        ctypes.c_float(eps), ctypes.c_float(-1.0 if momentum is None else momentum), ops._optptr(rm),
                                                        ops._optptr(rv), ops._optptr(nbt), 1 if relu else 0, _ptr(ws),
                                                        _ptr(counters), _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(y), _stream()),
                       "etch_bn_train_forward")
        else:
            rstd = torch.rsqrt(var + eps)
            scale = (g * rstd).contiguous()
            _lib.check(_lib.lib().etch_bn_apply(_c_long(R), C, _ptr(x), _c_long(x.stride(0)), _ptr(mean.contiguous()), _ptr(scale),
                                                _ptr(beta.detach().contiguous()), 1 if relu else 0, _ptr(y), _stream()), "etch_bn_apply")
        ctx.save_for_backward(x, y if relu else None, mean, rstd, g)
        ctx.relu, ctx.train = bool(relu), bool(train)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd, g = ctx.saved_tensors
        dy = dy.contiguous()
        R, C = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
        ws, counters = _ws(C, x.device), A.reduce_counters(x.device)
        _lib.check(_lib.lib().etch_bn_backward_fused(_c_long(R), C, _ptr(x), _c_long(x.stride(0)), ops._optptr(y), _ptr(dy), _ptr(mean), _ptr(rstd), _ptr(g),
                                                     1 if ctx.relu else 0, 1 if ctx.train else 0, _ptr(ws), _ptr(counters),
                                                     ops._optptr(dx), _ptr(dgb[0]), _ptr(dgb[1]), _stream()), "etch_bn_backward_fused")
        return dx, dgb[0], dgb[1], None, None, None, None, None, None


def batch_norm(x2, m, relu=False):
    """torch.nn.BatchNorm1d `m` on rows x2 (R,C) (+ ReLU): batch statistics and running-statistic update in train() mode
    (momentum / unbiased variance / num_batches_tracked as torch.nn.functional.batch_norm), running statistics in eval() mode."""
    if m.training:
        track = m.track_running_stats and m.running_mean is not None
        running = (m.running_mean, m.running_var, m.num_batches_tracked, m.momentum) if track else (None, None, None, 0.0)
        return BatchNormFunction.apply(x2, m.weight, m.bias, None, None, m.eps, relu, True, running)
    return BatchNormFunction.apply(x2, m.weight, m.bias, m.running_mean.detach(), m.running_var.detach(), m.eps, relu, False, None)


def _segments(index, nseg):
    """(perm, seg) of a flat int index list for etch_segment_sum_rows, memoised on the index tensor (one sort per kNN table, shared by every
    gather of every block of a level)."""
    hit = getattr(index, "_etch_segments", None)
    if hit is None or hit[0] != (nseg, index._version):
        key, perm = torch.sort(index.reshape(-1).long(), stable=True)
        seg = torch.searchsorted(key, torch.arange(nseg + 1, device=key.device, dtype=torch.int64)).contiguous()
        hit = ((nseg, index._version), perm.contiguous(), seg)
        index._etch_segments = hit
    return hit[1], hit[2]


def _segment_sum(src, index, nseg):
    perm, seg = _segments(index, nseg)
    dst = torch.empty((nseg, src.shape[1]), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().etch_segment_sum_rows(_c_long(nseg), src.shape[1], _ptr(src), _ptr(perm), _ptr(seg), _ptr(dst), _stream()),
               "etch_segment_sum_rows")
    return dst


class GatherRowsFunction(torch.autograd.Function):
    """x (n,c) -> x[idx.view(-1)] (E,c) for an int32 index tensor (pointops.py:79-100 grouping); backward = ordered segment sum."""

    @staticmethod
    def forward(ctx, x, idx):
        x = x.contiguous()
        ctx.idx, ctx.n = idx, x.shape[0]
        return ops.gather_rows(x, idx.view(-1))

    @staticmethod
    def backward(ctx, dy):
        return _segment_sum(dy.contiguous(), ctx.idx, ctx.n), None


def gather_rows(x, idx):
    """idx: the (memoised) index tensor itself, any shape -- the segment table of its backward is cached on it."""
    return GatherRowsFunction.apply(x, idx)


class RowsMaxPoolFunction(torch.autograd.Function):
    """nn.MaxPool1d(ns) over consecutive groups of ns rows (pointtransformer_seg.py:66): y (m*ns,c) -> (m,c)."""

    @staticmethod
    def forward(ctx, y, ns):
        y = y.contiguous()
        ctx.save_for_backward(y)
        ctx.ns = ns
        return ops.rows_maxpool(y, ns)

    @staticmethod
    def backward(ctx, dout):
        (y,) = ctx.saved_tensors
        m, c = y.shape[0] // ctx.ns, y.shape[1]
        dy = torch.empty_like(y)
        _lib.check(_lib.lib().etch_rows_maxpool_backward(_c_long(m), ctx.ns, c, _ptr(y), _ptr(dout.contiguous()), _ptr(dy), _stream()),
                   "etch_rows_maxpool_backward")
        return dy, None


class SoftmaxAggFunction(torch.autograd.Function):
    """softmax over the ns neighbours of logit (n*ns, cs), then out[i, s*cs + j] = sum_k sm[i,k,j] v[i,k,s*cs + j] (pointtransformer_seg.py:34-36)."""

    @staticmethod
    def forward(ctx, logit, v, ns):
        logit, v = logit.contiguous(), v.contiguous()
        E, cs = logit.shape
        c = v.shape[1]
        n = E // ns
        sm = torch.empty_like(logit)
        out = torch.empty((n, c), dtype=torch.float32, device=v.device)
        _lib.check(_lib.lib().etch_pt_softmax_agg(_c_long(n), ns, c, cs, _ptr(logit), _ptr(v), _ptr(sm), _ptr(out), _stream()), "etch_pt_softmax_agg")
        ctx.save_for_backward(sm, v)
        ctx.ns = ns
        return out

    @staticmethod
    def backward(ctx, dout):
        sm, v = ctx.saved_tensors
        E, cs = sm.shape
        c = v.shape[1]
        dlogit, dv = torch.empty_like(sm), torch.empty_like(v)
        _lib.check(_lib.lib().etch_pt_softmax_agg_backward(_c_long(E // ctx.ns), ctx.ns, c, cs, _ptr(sm), _ptr(v), _ptr(dout.contiguous()), _ptr(dlogit),
                                                           _ptr(dv), _stream()), "etch_pt_softmax_agg_backward")
        return dlogit, dv, None


class InterpolationFunction(torch.autograd.Function):
    """pointops.interpolation (pointops.py:164-178): out[i] = sum_k feat[idx[i,k]] w[i,k]; backward = weighted ordered segment sum."""

    @staticmethod
    def forward(ctx, feat, idx, dist):
        feat = feat.contiguous()
        zero = torch.zeros((idx.shape[0], feat.shape[1]), dtype=torch.float32, device=feat.device)
        out = ops.pt_interp_add(zero, feat, idx, dist)
        r = 1.0 / (dist + 1e-8)
        w = (r / ((r[:, 0] + r[:, 1]) + r[:, 2]).unsqueeze(1)).contiguous()        # the kernel's own weights (same operation order)
        ctx.save_for_backward(w)
        ctx.idx, ctx.m = idx, feat.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (w,) = ctx.saved_tensors
        perm, seg = _segments(ctx.idx, ctx.m)
        dout = dout.contiguous()
        dfe = torch.empty((ctx.m, dout.shape[1]), dtype=torch.float32, device=dout.device)
        _lib.check(_lib.lib().etch_weighted_segment_sum_rows(_c_long(ctx.m), dout.shape[1], 3, _ptr(dout), _ptr(w), _ptr(perm), _ptr(seg), _ptr(dfe),
                                                             _stream()), "etch_weighted_segment_sum_rows")
        return dfe, None, None


def _segment_ids(o, n):
    """Row -> scan index for cumulative offsets o, memoised on the offset tensor."""
    hit = getattr(o, "_etch_segment_ids", None)
    if hit is None or hit.shape[0] != n:
        oh = pointops.host_offsets(o)
        counts = torch.tensor([e - s for s, e in zip([0] + oh[:-1], oh)], dtype=torch.int64)
        hit = torch.repeat_interleave(torch.arange(len(oh), dtype=torch.int32), counts).to(o.device)
        o._etch_segment_ids = hit
        o._etch_counts = counts.to(torch.float32).to(o.device)
    return hit, o._etch_counts


class SegMeanFunction(torch.autograd.Function):
    """Per-scan mean of the rows (pointtransformer_seg.py:85-88)."""

    @staticmethod
    def forward(ctx, x, o):
        x = x.contiguous()
        ctx.o, ctx.n = o, x.shape[0]
        return ops.seg_mean(x, o, o.shape[0])

    @staticmethod
    def backward(ctx, dg):
        ids, counts = _segment_ids(ctx.o, ctx.n)
        return ops.gather_rows((dg / counts.unsqueeze(1)).contiguous(), ids), None


class ConcatBcastFunction(torch.autograd.Function):
    """[x | g[scan of the row]] (pointtransformer_seg.py:88-89)."""

    @staticmethod
    def forward(ctx, x, g, o):
        x, g = x.contiguous(), g.contiguous()
        ctx.o = o
        return ops.concat_bcast(x, g, o, o.shape[0])

    @staticmethod
    def backward(ctx, dxc):
        c = dxc.shape[1] // 2
        n = dxc.shape[0]
        ids, _ = _segment_ids(ctx.o, n)
        dg = _segment_sum(dxc[:, c:].contiguous(), ids, ctx.o.shape[0])
        return dxc[:, :c].contiguous(), dg, None


def _rel(p, new_p, idx):
    """p[idx] - new_p (E,3): coordinates only, no gradient; one tensor per kNN table inside a pointops.knn_scope (every block of a level, in both nets,
    asks for the same one: 44 x 3 launches per training step were 12 x 3)."""
    def make():
        with torch.no_grad():
            m, ns = idx.shape
            return (p[idx.view(-1).long()].view(m, ns, 3) - new_p.view(m, 1, 3)).reshape(m * ns, 3).contiguous()
    return pointops._memo(("pt_rel", p.data_ptr(), new_p.data_ptr(), idx.data_ptr()), (p, new_p, idx), make)


class NeighbourDiffFunction(torch.autograd.Function):
    """w[i*ns + j] = gk[i*ns + j] - xq[i] + pr[i*ns + j] (pointtransformer_seg.py:32) as one autograd node: 2 launches forward and 2 backward (the
    broadcast's gradient is one row-group sum) where the element-wise form spent ~8 engine launches per layer."""

    @staticmethod
    def forward(ctx, gk, xq, pr, ns):
        n, c = xq.shape
        ctx.ns = ns
        return ((gk.view(n, ns, c) - xq.view(n, 1, c)) + pr.view(n, ns, c)).view(n * ns, c)

    @staticmethod
    def backward(ctx, dw):
        ns = ctx.ns
        c = dw.shape[1]
        dxq = dw.view(-1, ns, c).sum(1).neg_() if ctx.needs_input_grad[1] else None
        return dw, dxq, dw, None


def _lin(x, layer, act=None):
    return A.linear(x, layer.weight, layer.bias, act=act)


# ------------------------------------------------------------------------------------------------ the modules, un-fused
def pt_layer(m, p, x, o):
    """PointTransformerLayer.forward (pointtransformer_seg.py:25-37)."""
    n, c, ns = x.shape[0], m.out_planes, m.nsample
    xq, xk, xv = _lin(x, m.linear_q), _lin(x, m.linear_k), _lin(x, m.linear_v)
    idx = pointops.knnquery(ns, p, p, o, o)[0]
    rel = _rel(p, p, idx)                                                          # (E,3)
    lp = m.linear_p
    pr = A.linear(rel, lp[0].weight, lp[0].bias)                                    # Linear(3,3)
    pr = batch_norm(pr, lp[1], relu=True)
    pr = A.linear(pr, lp[3].weight, lp[3].bias)                                     # Linear(3,c) (K = 3: the dense kernels take any K; the pads were launches)
    gk, gv = gather_rows(xk, idx), gather_rows(xv, idx)
    w = NeighbourDiffFunction.apply(gk, xq, pr, ns)                                 # x_k - x_q + p_r (pointtransformer_seg.py:32)
    lw = m.linear_w
    w = batch_norm(w, lw[0], relu=True)
    w = batch_norm(_lin(w, lw[2]), lw[3], relu=True)
    w = _lin(w, lw[5])
    return SoftmaxAggFunction.apply(w, gv + pr, ns)


def pt_block(m, p, x, o):
    """PointTransformerBlock.forward (pointtransformer_seg.py:113-122)."""
    y = batch_norm(A.linear(x, m.linear1.weight), m.bn1, relu=True)
    y = batch_norm(pt_layer(m.transformer2, p, y, o), m.bn2, relu=True)
    y = batch_norm(A.linear(y, m.linear3.weight), m.bn3)
    return torch.relu(y + x)


def transition_down(m, p, x, o):
    """TransitionDown.forward (pointtransformer_seg.py:52-68).  stride != 1: W [p_j - p_i | x_j] = Wp (p_j - p_i) + Wx x_j with the feature part
    evaluated per SOURCE point and gathered (the same sums as the grouped rows of the reference)."""
    w = m.linear.weight
    if m.stride != 1:
        oh = pointops.host_offsets(o)
        from .models.pointtransformer_seg import downsampled_offsets
        n_o = pointops.make_offsets(downsampled_offsets(oh, m.stride), p.device)
        idx = pointops.furthestsampling(p, o, n_o)
        n_p = pointops.gather_rows(p, idx)
        kidx = pointops.knnquery(m.nsample, p, n_p, o, n_o)[0]
        ux = A.linear(x, w[:, 3:])
        z = gather_rows(ux, kidx) + A.linear(_rel(p, n_p, kidx), w[:, :3])
        y = batch_norm(z, m.bn, relu=True)
        return n_p, RowsMaxPoolFunction.apply(y, m.nsample), n_o
    if x.shape[1] != w.shape[1]:                                                    # zero-padded input columns (see unet)
        w = F.pad(w, (0, x.shape[1] - w.shape[1]))
    return p, batch_norm(A.linear(x, w), m.bn, relu=True), o


def transition_up(m, pxo1, pxo2=None):
    """TransitionUp.forward (pointtransformer_seg.py:81-98)."""
    if pxo2 is None:
        _, x, o = pxo1
        g = _lin(SegMeanFunction.apply(x, o), m.linear2[0], act="relu")
        xc = ConcatBcastFunction.apply(x, g, o)
        return batch_norm(_lin(xc, m.linear1[0]), m.linear1[1], relu=True)
    p1, x1, o1 = pxo1
    p2, x2, o2 = pxo2
    a = batch_norm(_lin(x1, m.linear1[0]), m.linear1[1], relu=True)
    b = batch_norm(_lin(x2, m.linear2[0]), m.linear2[1], relu=True)
    idx, dist = pointops.knnquery(3, p2, p1, o2, o1)[:2]
    return a + InterpolationFunction.apply(b, idx, dist)


def _level(seq, first, rest_args):
    p, x, o = first
    for blk in list(seq)[1:]:
        x = pt_block(blk, p, x, o)
    return p, x, o


def unet(net, pxo):
    """_PointTransformerBase._unet (pointtransformer_seg.py:163-178, 237-252)."""
    p0, x0, o0 = pxo
    if net.c != 3:
        pad = (-(p0.shape[1] + x0.shape[1])) % 4
        x0 = torch.cat((p0, x0) + ((x0.new_zeros((x0.shape[0], pad)),) if pad else ()), 1)
    else:
        x0 = p0
    levels = []
    p, x, o = p0, x0, o0
    for enc in (net.enc1, net.enc2, net.enc3, net.enc4, net.enc5):
        p, x, o = _level(enc, transition_down(enc[0], p, x, o), None)
        levels.append([p, x, o])
    p5, x5, o5 = levels[4]
    levels[4][1] = _level(net.dec5, (p5, transition_up(net.dec5[0], [p5, x5, o5]), o5), None)[1]
    for li, dec in ((3, net.dec4), (2, net.dec3), (1, net.dec2), (0, net.dec1)):
        pl, xl, ol = levels[li]
        levels[li][1] = _level(dec, (pl, transition_up(dec[0], [pl, xl, ol], levels[li + 1]), ol), None)[1]
    return levels[0][1]


def confidence_forward(net, pxo):
    """PointTransformer_confidence.forward (pointtransformer_seg.py:163-195) -> (part_labels (B,N,k), confidences (B,N,1))."""
    p0, x0, o0 = pxo
    B = o0.shape[0]
    N = p0.shape[0] // B
    k = net.k
    with pointops.knn_scope():
        x1 = unet(net, pxo)                                                         # (B*N, 128)
    cls, confi = net.cls, net.confi
    h = batch_norm(A.linear(x1, cls[0].weight.view(cls[0].out_channels, -1), cls[0].bias), cls[1], relu=True)
    logits = A.linear(h, cls[3].weight.view(k, -1), cls[3].bias)                    # (B*N, k)
    hc = A.linear(x1, confi[0].weight.view(confi[0].out_channels, -1), confi[0].bias, act="relu")      # (B*N, 128 k)
    J = hc.shape[1] // k
    conf_k = (hc.view(-1, k, J) * confi[2].weight.view(1, k, J)).sum(-1) + confi[2].bias              # grouped Conv1d(128 k, k, groups = k)
    conf = (torch.softmax(logits, dim=1) * conf_k).sum(1, keepdim=True)
    return logits.view(B, N, k), conf.view(B, N, 1)


def magnitude_forward(net, pxo):
    """PointTransformer_magnitude.forward (pointtransformer_seg.py:237-260) -> (B,N,1)."""
    p0, x0, o0 = pxo
    B = o0.shape[0]
    N = p0.shape[0] // B
    with pointops.knn_scope():
        x1 = unet(net, pxo)
    fl = net.final_layer
    h = batch_norm(_lin(x1, fl[0]), fl[1], relu=True)
    y = (h * fl[3].weight.view(1, -1)).sum(1, keepdim=True) + fl[3].bias            # Linear(64, 1)
    return y.view(B, N, 1)
