"""MI355X counterpart of /root/reference/src/models/pointtransformer_seg.py: identical module tree and
state-dict keys; Linear(+BN+ReLU) layers run as one fp32-MFMA GEMM with a fused epilogue, the kNN
vector-attention core is one kernel per layer, FPS / kNN are the HIP index kernels."""
import os

import torch
import torch.nn as nn

from .. import ops
from ..vgtk_so3conv import _Derived
from . import pointops


def fold_bn(bn):
    """eval-mode BatchNorm1d -> (scale, shift): y = x*scale + shift."""
    s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    return s.contiguous(), (bn.bias.detach() - bn.running_mean * s).contiguous()



def downsampled_offsets(oh, stride):
    """TransitionDown's per-scan point counts (pointtransformer_seg.py:55-59: count += (o[i] - o[i-1]) // stride) as cumulative offsets.
    A scan that runs out of points before the last level (fewer than 256 points for the 4 x stride-4 nets) leaves the reference with
    empty segments and undefined neighbour queries; here it is an error instead of an out-of-bounds read on the GPU."""
    n_o, count, prev = [], 0, 0
    for i, end in enumerate(oh):
        m = (end - prev) // stride
        if m <= 0:
            raise ValueError(f"scan {i} has {end - prev} points at a stride-{stride} TransitionDown: too few for the five-level "
                             f"Point-Transformer nets (a scan needs at least 256 points)")
        count += m
        n_o.append(count)
        prev = end
    return n_o

class PointTransformerLayer(nn.Module):
    """pointtransformer_seg.py:8-37."""

    def __init__(self, in_planes, out_planes, share_planes=8, nsample=16):
        super().__init__()
        self.mid_planes = mid_planes = out_planes // 1
        self.out_planes, self.share_planes, self.nsample = out_planes, share_planes, nsample
        assert share_planes == 8
        self.linear_q = nn.Linear(in_planes, mid_planes)
        self.linear_k = nn.Linear(in_planes, mid_planes)
        self.linear_v = nn.Linear(in_planes, out_planes)
        self.linear_p = nn.Sequential(nn.Linear(3, 3), nn.BatchNorm1d(3), nn.ReLU(inplace=True), nn.Linear(3, out_planes))
        self.linear_w = nn.Sequential(nn.BatchNorm1d(mid_planes), nn.ReLU(inplace=True), nn.Linear(mid_planes, mid_planes // share_planes),
                                      nn.BatchNorm1d(mid_planes // share_planes), nn.ReLU(inplace=True),
                                      nn.Linear(out_planes // share_planes, out_planes // share_planes))
        self.softmax = nn.Softmax(dim=1)
        self._d = _Derived()

    def _apply(self, fn, *a, **k):
        self.__dict__["_slots"] = None        # .to() / .cuda() / .float(): the tensors behind the cached slots may be replaced
        return super()._apply(fn, *a, **k)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            self.__dict__["_slots"] = None    # a sub-module or Parameter object was re-assigned
        super().__setattr__(name, value)

    def _load_from_state_dict(self, *a, **k):
        self.__dict__["_slots"] = None        # load_state_dict(assign=True) replaces the Parameter objects
        return super()._load_from_state_dict(*a, **k)

    def _derived(self):
        # the tensors the derived constants depend on.  Module.parameters() / buffers() walk the module tree (named_modules), and 72 such walks per forward
        # were 45 % of the host's time per step (cProfile on the GPU box): the tree is walked ONCE into a list of (owner's dict, name) slots and every call
        # reads the CURRENT tensor of each slot (a list comprehension over ~30 dict lookups).  In-place updates (optimizer steps, load_state_dict) keep the
        # objects and are seen by _Derived's (data_ptr, version) key; a re-assigned Parameter (module.weight = nn.Parameter(...), load_state_dict(
        # assign=True)) is seen because the slot is looked up again; a REPLACED sub-module invalidates the slots (__setattr__ here; replacing a
        # module two levels down, e.g. layer.linear_p[0] = ..., needs `layer._slots = None`)
        slots = self.__dict__.get("_slots")
        if slots is None:
            slots = [(m._parameters, n) for m in self.modules() for n in m._parameters] + [(m._buffers, n) for m in self.modules() for n in m._buffers]
            self.__dict__["_slots"] = slots
        ps = [t for t in (d[n] for d, n in slots) if t is not None]

        def build():
            d = lambda t: t.detach().contiguous()
            wqkv = torch.cat([d(self.linear_q.weight), d(self.linear_k.weight), d(self.linear_v.weight)], 0).contiguous()
            bqkv = torch.cat([d(self.linear_q.bias), d(self.linear_k.bias), d(self.linear_v.bias)], 0).contiguous()
            sp, tp = fold_bn(self.linear_p[1])
            s0, t0 = fold_bn(self.linear_w[0])
            s3, t3 = fold_bn(self.linear_w[3])
            params = [d(self.linear_p[0].weight), d(self.linear_p[0].bias), sp, tp, d(self.linear_p[3].weight), d(self.linear_p[3].bias),
                      s0, t0, d(self.linear_w[2].weight).t().contiguous(), d(self.linear_w[2].bias), s3, t3,
                      d(self.linear_w[5].weight), d(self.linear_w[5].bias)]
            return wqkv, bqkv, params, (d(self.linear_w[2].weight), d(self.linear_w[2].bias), s3, t3, d(self.linear_w[5].weight), d(self.linear_w[5].bias))

        return self._d.get(ps, build)

    def forward(self, pxo, out_bn=None) -> torch.Tensor:
        p, x, o = pxo
        wqkv, bqkv, params, mlp = self._derived()
        qkv = ops.linear(x, wqkv, bias=bqkv)
        idx = pointops.knnquery(self.nsample, p, p, o, o)[0]
        allp = params + list(out_bn if out_bn is not None else (None, None))
        if self.attention_impl == "valu":
            return ops.pt_attention(p, qkv, self.out_planes, idx, allp, self.nsample)
        if self.attention_impl == "mfma" and (self.out_planes, self.nsample) in ops.PT_MFMA_SHAPES:
            return ops.pt_attention_mfma(p, qkv, self.out_planes, idx, allp, self.nsample, mlp[0])
        return ops.pt_attention_split(p, qkv, self.out_planes, idx, allp, self.nsample, *mlp)

    # "mfma": the whole attention core in one matrix-core kernel (etch_pt_attention_mfma); "split": prep kernel -> 2 x etch_linear ->
    # aggregate kernel (also the path of shapes the fused kernel is not instantiated for); "valu": the single-kernel VALU variant
    attention_impl = "mfma"


class TransitionDown(nn.Module):
    """pointtransformer_seg.py:40-68."""
    split_linear = True    # stride != 1: per-source-point GEMM + gather-max kernel instead of grouped rows -> GEMM -> maxpool

    def __init__(self, in_planes, out_planes, stride=1, nsample=16):
        super().__init__()
        self.stride, self.nsample = stride, nsample
        if stride != 1:
            self.linear = nn.Linear(3 + in_planes, out_planes, bias=False)
            self.pool = nn.MaxPool1d(nsample)
        else:
            self.linear = nn.Linear(in_planes, out_planes, bias=False)
        self.bn = nn.BatchNorm1d(out_planes)
        self.relu = nn.ReLU(inplace=True)
        self._d, self._dw, self._ds = _Derived(), _Derived(), _Derived()

    def forward(self, pxo):
        p, x, o = pxo
        s, t = self._d.get([self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var], lambda: fold_bn(self.bn))
        w = self.linear.weight.detach()
        if self.stride != 1:
            oh = pointops.host_offsets(o)
            n_o = downsampled_offsets(oh, self.stride)
            n_o_t = pointops.make_offsets(n_o, p.device, like=(o, self.stride))
            idx = pointops.furthestsampling(p, o, n_o_t)
            n_p = pointops.gather_rows(p, idx)
            kidx = pointops.knnquery(self.nsample, p, n_p, o, n_o_t)[0]
            if self.split_linear and w.shape[0] % 4 == 0:
                # W [p_j - p_i | x_j] = Wp (p_j - p_i) + Wx x_j: the feature part is one GEMM over the n SOURCE points (not the
                # m*ns = 4n grouped rows); coordinate part, BN, ReLU and the neighbour max in one gather kernel
                wpx = self._ds.get([self.linear.weight], lambda: (w[:, :3].contiguous(), w[:, 3:].contiguous()))
                ux = ops.linear(x, wpx[1])
                x = ops.pt_down_gather_max(ux, p, n_p, kidx, wpx[0], s, t)
            else:
                # rows and weight padded to a multiple of 4 columns with zeros (3 + c is odd): same sums, 16-byte load path in the GEMM
                g = ops.pt_group(p, n_p, x, kidx, pad_to=4)                    # (m*ns, 3+c (+pad))
                wp = self._dw.get([self.linear.weight], lambda: torch.nn.functional.pad(w, (0, g.shape[1] - w.shape[1])).contiguous())
                y = ops.linear(g, wp, scale=s, shift=t, act="relu")            # Linear -> BN -> ReLU
                x = ops.rows_maxpool(y, self.nsample)
            p, o = n_p, n_o_t
        else:
            if x.shape[1] != w.shape[1]:                                       # zero-padded input columns (see _unet)
                w = self._dw.get([self.linear.weight], lambda: torch.nn.functional.pad(w, (0, x.shape[1] - w.shape[1])).contiguous())
            x = ops.linear(x, w, scale=s, shift=t, act="relu")
        return [p, x, o]


class TransitionUp(nn.Module):
    """pointtransformer_seg.py:71-98."""

    def __init__(self, in_planes, out_planes=None):
        super().__init__()
        if out_planes is None:
            self.linear1 = nn.Sequential(nn.Linear(2 * in_planes, in_planes), nn.BatchNorm1d(in_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, in_planes), nn.ReLU(inplace=True))
        else:
            self.linear1 = nn.Sequential(nn.Linear(out_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))
        self._d = _Derived()

    def _bn(self):
        bns = [m for m in (self.linear1[1], self.linear2[1] if len(self.linear2) > 2 else None) if m is not None]
        ts = [t for m in bns for t in (m.weight, m.bias, m.running_mean, m.running_var)]
        return self._d.get(ts, lambda: [fold_bn(m) for m in bns])

    def forward(self, pxo1, pxo2=None):
        folded = self._bn()
        l1 = self.linear1[0]
        if pxo2 is None:
            _, x, o = pxo1
            nseg = o.shape[0]
            g = ops.seg_mean(x, o, nseg)
            g = ops.linear(g, self.linear2[0].weight.detach(), bias=self.linear2[0].bias.detach(), act="relu")
            xc = ops.concat_bcast(x, g, o, nseg)
            s, t = folded[0]
            return ops.linear(xc, l1.weight.detach(), bias=l1.bias.detach(), scale=s, shift=t, act="relu")
        p1, x1, o1 = pxo1
        p2, x2, o2 = pxo2
        (s1, t1), (s2, t2) = folded
        a = ops.linear(x1, l1.weight.detach(), bias=l1.bias.detach(), scale=s1, shift=t1, act="relu")
        b = ops.linear(x2, self.linear2[0].weight.detach(), bias=self.linear2[0].bias.detach(), scale=s2, shift=t2, act="relu")
        return pointops.interpolation(p2, p1, b, o2, o1, add_to=a)


class PointTransformerBlock(nn.Module):
    """pointtransformer_seg.py:101-122."""
    expansion = 1

    def __init__(self, in_planes, planes, share_planes=8, nsample=16):
        super().__init__()
        self.linear1 = nn.Linear(in_planes, planes, bias=False)
        self.bn1 = nn.BatchNorm1d(planes)
        self.transformer2 = PointTransformerLayer(planes, planes, share_planes, nsample)
        self.bn2 = nn.BatchNorm1d(planes)
        self.linear3 = nn.Linear(planes, planes * self.expansion, bias=False)
        self.bn3 = nn.BatchNorm1d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self._d = _Derived()

    # run_blocks: True = the K1 / K2 fused kernels (2 launches per block, blocks of a level chained: 1 + b launches per level); False = 4
    # kernels per block.  Default OFF: measured on the bench workload the fused form is 1.0 - 3.4x SLOWER per level (a 16-point tile
    # re-streams every c x c weight from L2 and its GEMM chain is latency-bound; profiles/r03_pt_block_fusion.txt).  ETCH_PT_FUSED=1 enables it.
    fused = os.environ.get("ETCH_PT_FUSED", "0") == "1"       # (needs a library built with ETCH_BUILD_EXPERIMENTS=1: ops.pt_block_k1 raises otherwise)

    def _fused_params(self):
        """Argument sets of the fused block kernels (etch_pt_block_k1 / _k2), rebuilt when a parameter changes."""
        bns = (self.bn1, self.bn2, self.bn3)
        f = self._d.get([t for m in bns for t in (m.weight, m.bias, m.running_mean, m.running_var)], lambda: [fold_bn(m) for m in bns])
        wqkv, bqkv, params, mlp = self.transformer2._derived()
        return dict(k1=(self.linear1.weight.detach(), f[0][0], f[0][1], wqkv, bqkv), attn=params + [f[1][0], f[1][1]], w2=mlp[0],
                    w3=self.linear3.weight.detach(), s3=f[2][0], t3=f[2][1])

    def forward(self, pxo):
        p, x, o = pxo
        bns = (self.bn1, self.bn2, self.bn3)
        f = self._d.get([t for m in bns for t in (m.weight, m.bias, m.running_mean, m.running_var)], lambda: [fold_bn(m) for m in bns])
        y = ops.linear(x, self.linear1.weight.detach(), scale=f[0][0], shift=f[0][1], act="relu")
        y = self.transformer2([p, y, o], out_bn=f[1])                                     # bn2 + relu fused into the kernel tail
        y = ops.linear(y, self.linear3.weight.detach(), scale=f[2][0], shift=f[2][1], res=x, res_mode=1, act="relu")
        return [p, y, o]


def run_blocks(blocks, pxo):
    """A run of consecutive PointTransformerBlocks of one level (pointtransformer_seg.py:101-122 each) with the fused block kernels:
    1 + len(blocks) launches (K1 of the first block; K2 of every block, carrying the next block's K1) instead of 4 per block.  Falls back
    to the blocks' own forward for shapes the fused kernels are not built for."""
    blocks = list(blocks)
    p, x, o = pxo
    if not blocks:
        return [p, x, o]
    b0 = blocks[0]
    c, ns = b0.transformer2.out_planes, b0.transformer2.nsample
    if not (PointTransformerBlock.fused and x.is_cuda and (c, ns) in ops.PT_BLOCK_SHAPES and x.shape[1] == c and
            all(b.transformer2.out_planes == c and b.transformer2.nsample == ns for b in blocks)):
        for b in blocks:
            p, x, o = b([p, x, o])
        return [p, x, o]
    idx = pointops.knnquery(ns, p, p, o, o)[0]
    der = [b._fused_params() for b in blocks]
    qkv = ops.pt_block_k1(x, *der[0]["k1"])
    for i, d in enumerate(der):
        nxt = der[i + 1]["k1"] if i + 1 < len(der) else None
        x, qkv = ops.pt_block_k2(p, qkv, c, idx, d["attn"], ns, d["w2"], d["w3"], d["s3"], d["t3"], x, next_k1=nxt)
    return [p, x, o]


class _PointTransformerBase(nn.Module):
    def _make_enc(self, block, planes, blocks, share_planes=8, stride=1, nsample=16):
        layers = [TransitionDown(self.in_planes, planes * block.expansion, stride, nsample)]
        self.in_planes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_planes, self.in_planes, share_planes, nsample=nsample))
        return nn.Sequential(*layers)

    def _make_dec(self, block, planes, blocks, share_planes=8, nsample=16, is_head=False):
        layers = [TransitionUp(self.in_planes, None if is_head else planes * block.expansion)]
        self.in_planes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_planes, self.in_planes, share_planes, nsample=nsample))
        return nn.Sequential(*layers)

    def _build(self, block, blocks, c, planes):
        self.in_planes = c
        share_planes = 8
        stride, nsample = [1, 4, 4, 4, 4], [8, 16, 16, 16, 16]
        for i in range(5):
            setattr(self, f"enc{i + 1}", self._make_enc(block, planes[i], blocks[i], share_planes, stride=stride[i], nsample=nsample[i]))
        self.dec5 = self._make_dec(block, planes[4], 2, share_planes, nsample=nsample[4], is_head=True)
        for i in (3, 2, 1, 0):
            setattr(self, f"dec{i + 1}", self._make_dec(block, planes[i], 2, share_planes, nsample=nsample[i]))

    def _unet(self, pxo):
        p0, x0, o0 = pxo
        if self.c != 3:
            # [p | x] zero-padded to a multiple of 4 columns (enc1's weight is padded to match): 16-byte load path of the GEMM
            pad = (-(p0.shape[1] + x0.shape[1])) % 4
            x0 = torch.cat((p0, x0) + ((x0.new_zeros((x0.shape[0], pad)),) if pad else ()), 1)
        else:
            x0 = p0
        enc = lambda m, pxo_: run_blocks(m[1:], m[0](pxo_))            # TransitionDown, then the level's blocks (fused kernels)
        p1, x1, o1 = enc(self.enc1, [p0, x0, o0])
        p2, x2, o2 = enc(self.enc2, [p1, x1, o1])
        p3, x3, o3 = enc(self.enc3, [p2, x2, o2])
        p4, x4, o4 = enc(self.enc4, [p3, x3, o3])
        p5, x5, o5 = enc(self.enc5, [p4, x4, o4])
        x5 = run_blocks(self.dec5[1:], [p5, self.dec5[0]([p5, x5, o5]), o5])[1]
        x4 = run_blocks(self.dec4[1:], [p4, self.dec4[0]([p4, x4, o4], [p5, x5, o5]), o4])[1]
        x3 = run_blocks(self.dec3[1:], [p3, self.dec3[0]([p3, x3, o3], [p4, x4, o4]), o3])[1]
        x2 = run_blocks(self.dec2[1:], [p2, self.dec2[0]([p2, x2, o2], [p3, x3, o3]), o2])[1]
        x1 = run_blocks(self.dec1[1:], [p1, self.dec1[0]([p1, x1, o1], [p2, x2, o2]), o1])[1]
        return x1


class PointTransformer_confidence(_PointTransformerBase):
    """pointtransformer_seg.py:125-195."""

    def __init__(self, block, blocks, c=6, k=13):
        super().__init__()
        self.c, self.k = c, k
        planes = [128, 128, 256, 256, 512]
        self._build(block, blocks, c, planes)
        self.cls = nn.Sequential(nn.Conv1d(planes[0], planes[0], 1), nn.BatchNorm1d(planes[0]), nn.ReLU(), nn.Conv1d(planes[0], k, 1))
        self.confi = nn.Sequential(nn.Conv1d(planes[0], planes[0] * k, 1), nn.ReLU(), nn.Conv1d(planes[0] * k, 1 * k, 1, groups=k))
        self._d, self._dw = _Derived(), _Derived()

    def forward(self, pxo):
        p0, x0, o0 = pxo
        B = o0.shape[0]
        N = p0.shape[0] // B
        with pointops.knn_scope():
            x1 = self._unet(pxo)                                                     # (B*N, 128)
        s, t = self._d.get([self.cls[1].weight, self.cls[1].bias, self.cls[1].running_mean, self.cls[1].running_var], lambda: fold_bn(self.cls[1]))
        h = ops.linear(x1, self.cls[0].weight.detach().view(self.cls[0].out_channels, -1), bias=self.cls[0].bias.detach(), scale=s, shift=t, act="relu")
        logits = ops.linear(h, self.cls[3].weight.detach().view(self.k, -1), bias=self.cls[3].bias.detach())          # (B*N, k)
        w0 = self.confi[0].weight.detach().view(self.confi[0].out_channels, -1)
        b0 = self.confi[0].bias.detach()
        w2 = self.confi[2].weight.detach().view(self.k, -1).contiguous()
        b2 = self.confi[2].bias.detach()
        # confi: Conv1d(128, 128k) -> ReLU -> grouped Conv1d(128k, k) fused; the (B*N, 128k) hidden layer is never materialised
        wp = self._dw.get([self.confi[0].weight], lambda: ops.permute_weight_frag_grouped(w0.contiguous(), w2.shape[1]))
        conf_k = ops.linear_relu_dot(x1, w0, b0, w2.view(-1), b2, self.k, wp=wp)
        conf = ops.softmax_dot(logits, conf_k)
        return logits.view(B, N, self.k), conf.view(B, N, 1)


class PointTransformer_magnitude(_PointTransformerBase):
    """pointtransformer_seg.py:199-260."""

    def __init__(self, block, blocks, c=6, k=1):
        super().__init__()
        self.c = c
        assert k == 1, "Please check output dim of PointTransformer_magnitude"
        planes = [64, 128, 256, 256, 512]
        self._build(block, blocks, c, planes)
        self.final_layer = nn.Sequential(nn.Linear(planes[0], planes[0]), nn.BatchNorm1d(planes[0]), nn.ReLU(inplace=True), nn.Linear(planes[0], 1))
        self._d = _Derived()

    def forward(self, pxo):
        p0, x0, o0 = pxo
        B = o0.shape[0]
        N = p0.shape[0] // B
        with pointops.knn_scope():
            x1 = self._unet(pxo)
        fl = self.final_layer
        s, t = self._d.get([fl[1].weight, fl[1].bias, fl[1].running_mean, fl[1].running_var], lambda: fold_bn(fl[1]))
        h = ops.linear(x1, fl[0].weight.detach(), bias=fl[0].bias.detach(), scale=s, shift=t, act="relu")
        y = ops.linear(h, fl[3].weight.detach(), bias=fl[3].bias.detach())           # (B*N, 1)
        return y.view(B, N, 1)


def get_pointtransformer_confidence(**kwargs):
    return PointTransformer_confidence(PointTransformerBlock, [2, 3, 4, 6, 3], **kwargs)


def get_pointtransformer_magnitude(**kwargs):
    return PointTransformer_magnitude(PointTransformerBlock, [2, 3, 4, 6, 3], **kwargs)


def prefetch_indices(p0, o0, strides=(1, 4, 4, 4, 4), nsamples=(8, 16, 16, 16, 16)):
    """Issue every FPS / kNN query of the Point-Transformer U-Net for (p0, o0) -- they depend on the coordinates only --
    so that they land in the active pointops.knn_scope cache.  The model runs this on a side stream while the EPN
    encoder is busy; both nets then find their indices memoised.  Returns the tensors created (for record_stream)."""
    made = []
    levels = []
    p, o = p0, o0
    for li in range(5):
        if strides[li] != 1:
            oh = pointops.host_offsets(o)
            n_o = downsampled_offsets(oh, strides[li])
            n_o_t = pointops.make_offsets(n_o, p.device, like=(o, strides[li]))
            idx = pointops.furthestsampling(p, o, n_o_t)
            n_p = pointops.gather_rows(p, idx)
            made += [idx, n_p, n_o_t] + list(pointops.knnquery(nsamples[li], p, n_p, o, n_o_t))
            p, o = n_p, n_o_t
        made += list(pointops.knnquery(nsamples[li], p, p, o, o))
        levels.append((p, o))
    for li in range(3, -1, -1):                                  # TransitionUp interpolation: coarse -> fine 3-NN
        (pf, of), (pc, oc) = levels[li], levels[li + 1]
        made += list(pointops.knnquery(3, pc, pf, oc, of))
    return made
