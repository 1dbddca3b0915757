"""MI355X counterpart of /root/reference/src/models/models_pointcloud.py: `GT_network_equiv` with the same
constructor contract (`option.{output_folder, EPN_input_radius, EPN_layer_num, markerset, device}`), the same
module tree / state-dict keys (reference checkpoints load unchanged) and the same forward signature and
result dictionary.  Every op on the path is a HIP kernel of libetch_hip.so."""
import os

import torch
import torch.nn.init as init
from torch import nn

from .. import ops
from ..config.EPN_options import get_default_cfg
from .direction_backbones import BatchMLP, StackedMHSA
from . import pointops
from .pointnet2_utils import propagate_cl
from .pointtransformer_seg import get_pointtransformer_confidence, get_pointtransformer_magnitude
from .so3net import build_model


SUPPORTED_EPN_LAYER_NUMS = frozenset({1, 2, 3, 4})       # models_pointcloud.py:34-48: feature widths 32 / 64 / 128 / 256


class GT_network_equiv(nn.Module):
    def __init__(self, option=None):
        super().__init__()
        self.option = option
        model_setting_file = os.path.join(option.output_folder, "EPN_model_setting_json")
        EPN_cfg = get_default_cfg()
        EPN_cfg.model.search_radius = option.EPN_input_radius
        mlp_layers = [[32, 32], [64, 64], [128, 128], [256, 256]]
        strides_layers = [2, 2, 2, 2]
        self.standard_vector = torch.tensor([0, 0, 1], dtype=torch.float32, requires_grad=False)
        EPN_layer_n = option.EPN_layer_num
        if EPN_layer_n not in SUPPORTED_EPN_LAYER_NUMS:
            raise ValueError(f"EPN_layer_num must be one of {sorted(SUPPORTED_EPN_LAYER_NUMS)} (models_pointcloud.py:34-48), got {EPN_layer_n}")
        EPN_feat_dim = mlp_layers[EPN_layer_n - 1][0]
        os.makedirs(option.output_folder, exist_ok=True)
        self.encoder = build_model(EPN_cfg, mlps=mlp_layers[:EPN_layer_n], strides=strides_layers[:EPN_layer_n], to_file=model_setting_file)
        # direction
        self.direction_encoder = StackedMHSA(embedding_dim=EPN_feat_dim, value_dim=128, num_heads=8, num_layers=2)
        self.direction_predictor = BatchMLP(in_features=128, out_features=128)
        self.so3_reg = nn.Conv1d(128, 1, 1)
        # magnitude / confidence
        self.magnitude_encoder = get_pointtransformer_magnitude(c=EPN_feat_dim + 3, k=1)
        self.confidence_encoder = get_pointtransformer_confidence(c=EPN_feat_dim + 3, k=len(option.markerset))
        self._reset_parameters()

    def _reset_parameters(self):
        """models_pointcloud.py:72-77."""
        for p in self.parameters():
            if p.dim() > 1:
                init.xavier_uniform_(p)

    def encode(self, f):
        return self.encoder(f)

    def preprocess_data(self, xyz, features):
        """models_pointcloud.py:82-92."""
        B, N, C = features.shape
        p = xyz.reshape(-1, 3)
        x = features.reshape(-1, C)
        oh = [N * (i + 1) for i in range(B)]
        o = pointops.offsets_tensor(oh, p.device)
        return p, x, o

    def decode_confidence(self, inv_feat, xyz):
        return self.confidence_encoder(self.preprocess_data(xyz, inv_feat))

    def decode_magnitude(self, inv_feat, xyz):
        return self.magnitude_encoder(self.preprocess_data(xyz, inv_feat))

    fold_linear_chains = True   # exact algebra: head_combine(last MHSA) o net[0] and net[2] o so3_reg are linear o linear

    def _folded(self):
        """Consecutive linear maps of the direction head folded in fp64 on the host (no non-linearity sits between them):
             H     = relu(att @ (W1 Wc)^T + (W1 bc + b1))     head_combine of the last MHSA layer, then direction_predictor.net[0]
             anc_w = H @ (W2^T w_reg) + (b2 . w_reg + b_reg)   direction_predictor.net[2], then so3_reg (Conv1d 128 -> 1)
           which removes two 128x128 GEMMs over all B*N*60 tokens (models_pointcloud.py:115-117 computes the same function)."""
        last = self.direction_encoder.self_attention_layers[-1]
        n0, n2 = self.direction_predictor.net[0], self.direction_predictor.net[2]
        ps = [last.head_combine.weight, last.head_combine.bias, n0.weight, n0.bias, n2.weight, n2.bias, self.so3_reg.weight, self.so3_reg.bias]

        def build():
            d = lambda t: t.detach().double().cpu()
            Wc, bc, W1, b1, W2, b2 = d(ps[0]), d(ps[1]), d(ps[2]), d(ps[3]), d(ps[4]), d(ps[5])
            wr, br = d(ps[6]).view(-1), d(ps[7]).view(-1)
            dev = ps[0].device
            Wf = (W1 @ Wc).float().contiguous().to(dev)
            bf = (W1 @ bc + b1).float().contiguous().to(dev)
            v = (W2.t() @ wr).float().contiguous().to(dev)
            c = (b2 @ wr + br[0]).float().view(1).to(dev)
            Wfq = ops.dirtail_weight_split(Wf) if tuple(Wf.shape) == (128, 64) else None      # the fused tail: 64-wide tokens
            # [bf | v | c]: the fused tail's constants (etch_mhsa_layer_dirtail), with the hidden units' powers of two folded in
            tab = ops.dirtail_constants(bf, v, c, Wfq.wsc) if Wfq is not None else torch.cat([bf, v, c]).contiguous()
            return Wf, bf, v, c, ops.permute_weight_frag_grouped(Wf), Wfq, tab

        if not hasattr(self, "_fold_cache"):
            from ..vgtk_so3conv import _Derived
            self._fold_cache = _Derived()
        return self._fold_cache.get(ps, build)

    fuse_direction_interp = True   # the first MHSA layer forms the 3-NN interpolated tokens itself (they are never written out)
    fuse_direction_tail = os.environ.get("ETCH_DIR_TAIL", "1") != "0"   # the last MHSA layer carries the folded tail (round 5; 0: mhsa_layer mode 2 + linear_relu_dot)

    def _can_fuse_interp(self):
        layers = self.direction_encoder.self_attention_layers
        return (self.fuse_direction_interp and self.fold_linear_chains and len(layers) >= 2 and layers[0].embedding_dim == 64
                and layers[0].value_dim == 64)

    def anchor_weights(self, tokens, interp=None):
        """tokens [T, 60, C] -> anc_w [T, 60]: direction_encoder -> direction_predictor -> so3_reg (models_pointcloud.py:115-117).
        interp = (coarse tokens (B,S,60,C), idx (B,N,3), w (B,N,3), order) instead of `tokens`: the 3-NN interpolation of :181-183 is
        done inside the first attention layer's kernel; the interpolated tokens are never written out."""
        layers = self.direction_encoder.self_attention_layers
        if interp is not None:
            feats_cl, idx3, w3, order = interp
            l0 = layers[0]
            x = ops.mhsa_interp_layer(feats_cl, idx3, w3, l0.query_transform.weight.detach(), l0.key_transform.weight.detach(),
                                      l0.value_transform.weight.detach(), l0.head_combine.weight.detach(), l0.head_combine.bias.detach(), order=order)
            T = x.shape[0] // 60
            x = x.view(T, 60, l0.value_dim)
            first = 1
        else:
            T = tokens.shape[0]
            if not self.fold_linear_chains:
                x = self.direction_predictor(self.direction_encoder(tokens))
                return ops.rowdot(x.view(T * 60, -1), self.so3_reg.weight.detach().view(-1), float(self.so3_reg.bias.detach().cpu())).view(T, 60)
            x = tokens
            first = 0
        for layer in layers[first:-1]:
            x = layer(x, x, x, residual=True)
        last = layers[-1]
        Wf, bf, v, c, Wfp, Wfq, tab = self._folded()
        if self.fuse_direction_tail and last.embedding_dim == 64:
            # the last layer's heads AND relu(att Wf^T + bf) . v + c in one kernel: neither the attention output nor the hidden layer leaves the chip
            return ops.mhsa_layer_dirtail(x.reshape(T * 60, 64).contiguous(), last.query_transform.weight.detach(), last.key_transform.weight.detach(),
                                          last.value_transform.weight.detach(), Wfq, tab)
        att = last.heads(x.reshape(T * 60, last.embedding_dim))                       # concatenated heads; head_combine is folded into Wf
        # relu(att Wf^T + bf) . v + c in one kernel: the (T*60, 128) hidden layer stays on chip
        return ops.linear_relu_dot(att, Wf, bf, v, c, 1, wp=Wfp).view(T, 60)

    def decode_direction(self, equiv_feat, anchors, initial_vectors, tokens_cl=None, interp=None):
        """models_pointcloud.py:111-126.  equiv_feat [B, N, C, 60] (reference layout) or tokens_cl [B, N, 60, C], or `interp` (see
        anchor_weights)."""
        if interp is not None:
            B, N, na = interp[1].shape[0], interp[1].shape[1], interp[0].shape[2]
            anc_w = self.anchor_weights(None, interp=interp)
        else:
            if tokens_cl is None:
                tokens_cl = equiv_feat.permute(0, 1, 3, 2).contiguous()
            B, N, na, C = tokens_cl.shape
            anc_w = self.anchor_weights(tokens_cl.view(B * N, na, C))
        self.last_anc_w = anc_w.view(B, N, na)
        iv = initial_vectors.reshape(-1, 3)[0].tolist()
        assert iv == [0.0, 0.0, 1.0], "only direction_mode='standard_vector' is implemented (as in the reference, :198-208)"
        d, _, _ = ops.so3_mean_dir(anc_w, anchors.contiguous())
        return d.view(B, N, 3)

    # ------------------------------------------------------------------------------------------------ differentiable path
    # Which path forward() takes:
    #   differentiable = None  (default)  the path with autograd history (etch_amd.autograd / autograd_pt: hand-written backward kernels, un-fused,
    #                                     several times slower and far more memory than inference) ONLY in train() mode with gradients enabled --
    #                                     what train.py:61,77 does; eval() mode always takes the fused inference path (results without history),
    #                                     also when the caller forgot torch.no_grad()
    #   differentiable = True             always (eval() mode: BatchNorm on its running statistics)
    #   differentiable = False            never
    differentiable = None
    _warned_switch = False
    _warned_eval_grad = False

    def wants_grad(self):
        if self.differentiable is not None:
            return bool(self.differentiable) and torch.is_grad_enabled()
        return self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def differentiable_supported(self):
        """The differentiable direction head covers every encoder depth of the reference (EPN_layer_num 1 / 2 / 3 / 4: 32 / 64 / 128 / 256-dim
        tokens, models_pointcloud.py:34-48): the fused attention layer at 64, the un-fused chain + etch_mhsa_attention_backward_dim otherwise."""
        return all(l.embedding_dim in (32, 64, 128, 256) for l in self.direction_encoder.self_attention_layers)

    def encode_differentiable(self, hitpts):
        """The EPN encoder through etch_amd.autograd (hand-written backward kernels): what train.py:77-101 back-propagates through
        so3net.py:23-33 / so3conv.py:171-183.  -> (xyz (B,3,S), feats_cl (B,S,60,C) with autograd history, anchors)."""
        from .. import autograd as A
        from .so3conv import input_xyz
        xyz = input_xyz(hitpts)
        b, _, n = xyz.shape
        feats = torch.ones((b, n, 60, 1), dtype=torch.float32, device=hitpts.device)        # occupancy features (functional.py:70-89)
        anchors = None
        for block in self.encoder.backbone:
            for conv in block.blocks:
                ic, itc = conv.inter_conv.conv, conv.intra_conv.conv
                ball, sidx, new_xyz = ic.group(xyz)
                rk = ic.rotated_kernels()
                y = A.inter_so3conv(feats, ic.basic_conv.W, ic.basic_conv.bias, xyz, new_xyz, ball, rk, ic.sigma)
                y = A.instnorm_leaky_relu(y)
                z = A.instnorm_leaky_relu(A.intra_so3conv(y, itc.basic_conv.W, itc.basic_conv.bias, itc.intra_idx))
                skip = feats if conv.stride == 1 else A.GatherPointsFunction.apply(feats, sidx)
                sc = conv.skip_conv
                s = A.instnorm_leaky_relu(A.linear(skip, sc.weight.view(sc.out_channels, -1), sc.bias))
                feats, xyz, anchors = z + s, new_xyz, itc.anchors
        return xyz, feats, anchors

    def direction_differentiable(self, tokens, anchors, B, N):
        """direction_encoder -> direction_predictor -> so3_reg -> so3_mean -> R @ [0,0,1] (models_pointcloud.py:111-126) on the propagated
        tokens (B*N, 60, E), with autograd history; un-fused (no folded linear chains: every parameter receives its own gradient)."""
        from .. import autograd as A
        x = tokens
        layers = list(self.direction_encoder.self_attention_layers)
        for l in layers[:-1]:
            x = A.mhsa_layer(x, l.query_transform.weight, l.key_transform.weight, l.value_transform.weight, l.head_combine.weight,
                             l.head_combine.bias, residual=True)
        l = layers[-1]
        h = A.linear(A.mhsa_heads(x, l.query_transform.weight, l.key_transform.weight, l.value_transform.weight), l.head_combine.weight,
                     l.head_combine.bias)
        n0, n2 = self.direction_predictor.net[0], self.direction_predictor.net[2]
        h = A.linear(A.linear(h, n0.weight, n0.bias, act="relu"), n2.weight, n2.bias)
        aw = A.linear(h, self.so3_reg.weight.view(1, -1), self.so3_reg.bias).view(B * N, 60)
        self.last_anc_w = aw.view(B, N, 60)
        return A.so3_mean_dir(aw, anchors).view(B, N, 3)

    def forward_differentiable(self, hitpts, pred_items, direction_mode):
        """forward() with autograd history for every parameter (train.py:77-101): the encoder and the direction head through etch_amd.autograd,
        the two Point-Transformer nets through etch_amd.autograd_pt (train-mode BatchNorm on batch statistics when self.training).  The encoder
        and the 3-NN propagation run once and feed all three heads."""
        from .. import autograd as A
        from .. import autograd_pt as P
        if direction_mode != "standard_vector":
            raise AssertionError("Not implemented")
        B, N, _ = hitpts.shape
        # the index ops (FPS / ball queries of the encoder, FPS / kNN chain of the nets: functions of the coordinates only, no gradient) on the index
        # stream like the inference path's: the nets' chain runs beside the encoder, and the two serial FPS launches of a small batch share one
        idx_ready = None
        if self.overlap_index_ops and hitpts.is_cuda:
            epn_ready, idx_ready = self._prefetch_indices(hitpts.detach(), B, N, "confidence" in pred_items or "magnitude" in pred_items)
            torch.cuda.current_stream().wait_event(epn_ready)
        xyz, feats_cl, anchors = self.encode_differentiable(hitpts)
        idx3, w3 = ops.prop3nn(hitpts, xyz)
        tokens = A.prop_interp(feats_cl, idx3, w3)                                 # (B,N,60,C): models_pointcloud.py:181-183
        results = {}
        if idx_ready is not None:
            torch.cuda.current_stream().wait_event(idx_ready)
        if "confidence" in pred_items or "magnitude" in pred_items:
            inv = tokens.mean(2)                                                    # the anchor mean (:184)
            pxo = self.preprocess_data(hitpts, inv)
            if "confidence" in pred_items:
                results["part_labels"], results["confidences"] = P.confidence_forward(self.confidence_encoder, pxo)
            if "magnitude" in pred_items:
                results["magnitude"] = P.magnitude_forward(self.magnitude_encoder, pxo)
        if "direction" in pred_items:
            results["direction"] = self.direction_differentiable(tokens.view(B * N, 60, feats_cl.shape[-1]), anchors, B, N)
        selected_indexs = self._const(('sel', B, N, str(hitpts.device)), lambda: torch.arange(0, N, device=hitpts.device).repeat(B, 1)).unsqueeze(-1).expand(-1, -1, 3)
        return results, selected_indexs

    def _const(self, key, make):
        """Per-shape constants of a forward (the identity `selected_indexs`, the broadcast standard vector): built once per (shape, device) instead of with
        2 - 3 launches per forward.  Read-only by contract: the reference returns fresh tensors, callers that write into them must clone."""
        c = self.__dict__.setdefault("_const_cache", {})
        v = c.get(key)
        if v is None:
            if len(c) > 64:
                c.clear()
            with torch.no_grad():
                v = c[key] = make()
        return v

    def forward(self, hitpts, pred_items=["direction", "magnitude"], direction_mode="standard_vector"):
        """models_pointcloud.py:146-221.  See `differentiable` above for when the path with autograd history is taken."""
        B, N, _ = hitpts.size()
        hitpts = hitpts.contiguous()
        if self.wants_grad():
            if self.differentiable_supported():
                if self.differentiable is None and not GT_network_equiv._warned_switch:
                    GT_network_equiv._warned_switch = True
                    import warnings
                    warnings.warn("GT_network_equiv: train() mode with gradients enabled -> the differentiable (un-fused, slower) path; "
                                  "call model.eval() or set model.differentiable = False for inference", stacklevel=2)
                with pointops.knn_scope():
                    return self.forward_differentiable(hitpts, pred_items, direction_mode)
            if self.differentiable:
                raise NotImplementedError("the differentiable direction head is built for 32 / 64 / 128 / 256-dim tokens (EPN_layer_num 1 .. 4)")
            import warnings
            warnings.warn("GT_network_equiv: no differentiable path for this token width; running the inference path (results carry no "
                          "autograd history)", stacklevel=2)
        elif (self.differentiable is None and not self.training and torch.is_grad_enabled() and not GT_network_equiv._warned_eval_grad
              and (hitpts.requires_grad or any(p.requires_grad for p in self.parameters()))):
            # the reference back-propagates in eval() mode too (frozen-BatchNorm fine-tuning, input / test-time gradients); here eval() takes the fused
            # inference path, whose results carry no autograd history -- say so once instead of returning history-free tensors silently
            GT_network_equiv._warned_eval_grad = True
            import warnings
            warnings.warn("GT_network_equiv: eval() mode with gradients enabled -> the fused inference path (results carry NO autograd history); "
                          "set model.differentiable = True for gradients in eval() mode, or wrap the call in torch.no_grad()", stacklevel=2)
        with pointops.knn_scope():
            return self._forward(hitpts, pred_items, direction_mode, B, N)

    pt_index_stream = os.environ.get("ETCH_PT_INDEX_STREAM", "0") == "1"      # the Point-Transformer nets' index chain on its own stream (A/B)
    overlap_index_ops = True   # run every coordinate-only index op (EPN FPS / ball queries, PT FPS / kNN) on a side stream
    input_producer = None      # stream that produced `hitpts` if it is not the current one (set by a pipelined caller)

    keepalive = None     # a list set by a pipelined caller around forward(): cross-stream tensors are appended to it instead of being record_stream()-ed

    def _cross_stream(self, tensors, streams):
        """Tensors allocated on one stream and read on others.  Default: Tensor.record_stream (the caching allocator then defers the reuse of the block
        until the other streams have passed the point of the free -- at the price of an event per (block, stream) that EVERY later allocation polls:
        torch.empty cost ~45 us instead of ~5 on the pipeline's enqueue thread, the largest single item of its time per step).  A caller that retires
        whole batches (pipeline.py) hands in a list instead and drops it when the batch's last kernel has completed: no events, same safety."""
        if self.keepalive is not None:
            self.keepalive.extend(tensors)
            return
        for t in tensors:
            for st in streams:
                t.record_stream(st)

    def _prefetch_indices(self, hitpts, B, N, want_pt):
        """FPS / ball queries of the EPN encoder and all FPS / kNN queries of both Point-Transformer nets depend on the
        coordinates only: issue them on a side HIP stream; the layers find them memoised (pointops.knn_scope).  The side
        stream waits for the producer of `hitpts`, not for the work already queued on the current stream: under the
        2-deep pipeline the index ops of batch i+1 (latency-bound, 32 workgroups) run while batch i still computes.
        Returns (event after the EPN part, event after everything)."""
        from .pointtransformer_seg import prefetch_indices
        from .so3conv import input_xyz
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side_stream"):
            # (normal priority: as a high-priority queue it starves the stage-2 fit, see pipeline.py)
            from ..utils.cu_streams import make_stream
            self._side_stream = make_stream("side", priority=int(os.environ.get("ETCH_INDEX_STREAM_PRIORITY", "0")))
        side = self._side_stream
        side.wait_stream(self.input_producer if self.input_producer is not None else main)
        made = []
        self._cross_stream([hitpts], [side])
        with torch.cuda.stream(side):
            cur = input_xyz(hitpts)
            made.append(cur)
            od = self._input_order(hitpts)
            if od is not None:
                made.append(od)
            if want_pt and self.fps_pair and not self.pt_index_stream:
                made += self._fps_pair_ahead(hitpts, cur, B, N)
            for block in self.encoder.backbone:
                for conv in block.blocks:
                    ic = conv.inter_conv.conv
                    ball, sidx, cur = ic.group(cur)
                    made += [ball, sidx, cur]
                    od = ic.order(cur)
                    if od is not None:
                        made.append(od)
            epn_ready = torch.cuda.Event()
            epn_ready.record(side)
            # (the Point-Transformer nets' FPS / kNN chain stays on this stream behind the EPN part: a stream of its own -- so that the next batch's first
            # FPS does not queue behind this batch's kNN queries -- measured no better, 6 streams on 4 hardware queues)
            if want_pt and not self.pt_index_stream:
                oh = [N * (i + 1) for i in range(B)]
                o = pointops.offsets_tensor(oh, hitpts.device)
                made += prefetch_indices(hitpts.view(-1, 3), o)
            done = torch.cuda.Event()
            done.record(side)
        if want_pt and self.pt_index_stream:
            # A/B switch (ETCH_PT_INDEX_STREAM=1): the nets' FPS / kNN chain on a stream of its own, so that it runs NEXT TO the encoder's FPS
            # (one workgroup per scan each: 8 + 8 of 256 compute units on the dense 8-scan shard of configs[4]) instead of behind it
            if not hasattr(self, "_side_stream2"):
                from ..utils.cu_streams import make_stream
                self._side_stream2 = make_stream("side", priority=int(os.environ.get("ETCH_INDEX_STREAM_PRIORITY", "0")))
            side2 = self._side_stream2
            side2.wait_stream(self.input_producer if self.input_producer is not None else main)
            self._cross_stream([hitpts], [side2])
            with torch.cuda.stream(side2):
                oh = [N * (i + 1) for i in range(B)]
                o = pointops.offsets_tensor(oh, hitpts.device)
                made += prefetch_indices(hitpts.view(-1, 3), o)
                done = torch.cuda.Event()
                done.record(side2)
        users = [main] + (list(self._heads_streams()) if self.concurrent_heads else [])
        self._cross_stream(made, users)
        return epn_ready, done

    fps_pair = os.environ.get("ETCH_FPS_PAIR", "1") != "0"

    def _fps_pair_ahead(self, hitpts, xyz_b3n, B, N):
        """The encoder's FPS and the nets' first FPS level in one launch (ops.fps_pair), handed to their callers through ops._FPS_READY: on the dense 8-scan
        shard of configs[4] the two chains (27 + 11 ms, eight workgroups each) no longer queue behind each other on the index stream.  Only where both
        will really be asked for with exactly these arguments (first conv strided and not lazily sampled, split FPS not selected)."""
        from .pointtransformer_seg import downsampled_offsets
        ops._FPS_READY.clear()            # (a forward that raised half-way must not leave a result behind for a later tensor at the same address)
        ic = self.encoder.backbone[0].blocks[0].inter_conv.conv
        m = -(-N // ic.stride)
        if ic.stride <= 1 or ic.lazy_sample or m == N or ops.fps_split_default(B, N, hitpts.device) >= 2:
            return []
        oh = [N * (i + 1) for i in range(B)]
        o = pointops.offsets_tensor(oh, hitpts.device)
        n_o = downsampled_offsets(oh, 4)
        n_o_t = pointops.make_offsets(n_o, hitpts.device, like=(o, 4))
        packed = hitpts.view(-1, 3)
        idx_a, idx_b = ops.fps_pair(xyz_b3n, m, packed, o, n_o_t, n_o)
        ops._FPS_READY[("vgtk", xyz_b3n.data_ptr(), int(m))] = idx_a
        ops._FPS_READY[("pointops", packed.data_ptr(), n_o_t.data_ptr())] = idx_b
        return [idx_a, idx_b, n_o_t]

    def _input_order(self, hitpts):
        """Morton order of the input points of each scan (scheduling hint for the gather-heavy kernels), memoised per forward and
        issued by the index stream."""
        if not hitpts.is_cuda:
            return None
        from .so3conv import input_xyz
        xyz = input_xyz(hitpts)
        return pointops._memo(("input_order", xyz.data_ptr(), tuple(xyz.shape)), (xyz,), lambda: ops.spatial_order(xyz))

    def _heads_streams(self):
        if not hasattr(self, "_head_streams"):
            # the nets are long chains of small kernels (the critical path of this phase), the direction head on the current
            # stream is a few chip-wide kernels: high-priority queues let the chains go first whenever they have work
            prio = int(os.environ.get("ETCH_HEAD_STREAM_PRIORITY", "-1"))
            from ..utils.cu_streams import make_stream
            if os.environ.get("ETCH_HEAD_STREAMS", "2") == "1":     # A/B switch: both nets on ONE side stream (four streams in all = HIP's four hardware queues)
                one = make_stream("side", priority=prio)
                self._head_streams = (one, one)
            else:
                self._head_streams = (make_stream("side", priority=prio), make_stream("side", priority=prio))
        return self._head_streams

    def _forward(self, hitpts, pred_items, direction_mode, B, N):
        idx_ready = None
        if self.overlap_index_ops and hitpts.is_cuda:
            epn_ready, idx_ready = self._prefetch_indices(hitpts, B, N, "confidence" in pred_items or "magnitude" in pred_items)
            torch.cuda.current_stream().wait_event(epn_ready)
        r, sample_idx_lists = self.encode(hitpts)
        so3_anchors = r.anchors
        selected_indexs = self._const(('sel', B, N, str(hitpts.device)), lambda: torch.arange(0, N, device=hitpts.device).repeat(B, 1)).unsqueeze(-1).expand(-1, -1, 3)
        # 3-NN propagation of the [C*60] equivariant features to all N points + anchor mean (:181-184), channels-last
        order = self._input_order(hitpts)
        interp = None
        if hitpts.is_cuda and self._can_fuse_interp():
            # the anchor mean is linear too: mean over the anchors at the COARSE points, then the 64-wide 3-NN blend (the nets that read
            # it start ~1 ms earlier); the equivariant tokens themselves are blended inside the first attention layer
            idx3, w3 = ops.prop3nn(hitpts, r.xyz)
            Bc, S, na, C = r.feats_cl.shape
            cmean = ops.token_mean(r.feats_cl.view(Bc * S, na, C))
            _, point_inv_feat = ops.prop_interp(cmean.view(Bc, S, 1, C), idx3, w3, order=order)
            point_equiv_cl, interp = None, (r.feats_cl, idx3, w3, order)
        else:
            point_equiv_cl, point_inv_feat = propagate_cl(hitpts, r.xyz, r.feats_cl, order=order)
        results = {}
        if idx_ready is not None:
            torch.cuda.current_stream().wait_event(idx_ready)
        # both Point-Transformer nets share FPS / kNN indices (same points, same offsets) through the enclosing knn_scope
        self._heads(results, pred_items, direction_mode, hitpts, point_inv_feat, point_equiv_cl, so3_anchors, B, N,
                    indices_prefetched=idx_ready is not None, interp=interp)
        return results, selected_indexs

    concurrent_heads = True    # run the confidence and magnitude nets on their own HIP streams next to the direction head
    direction_first = os.environ.get("ETCH_DIRECTION_FIRST", "1") != "0"      # host enqueue order of the three heads (A/B switch)
    defer_join = False         # True (set by a pipelined caller around forward): leave `pending_join` events instead of joining
    pending_join = None

    def _heads(self, results, pred_items, direction_mode, hitpts, point_inv_feat, point_equiv_cl, so3_anchors, B, N, indices_prefetched=False,
               interp=None):
        # The three heads are independent given the encoder output.  With `concurrent_heads` the two Point-Transformer nets
        # (many small launches: a few dozen workgroups at the deep levels) run on side streams while the direction head keeps
        # the matrix cores busy on the current one.  Only when every index tensor was produced ahead of time: a tensor
        # memoised by one branch and read by another would otherwise cross streams unsynchronised.
        fork = self.concurrent_heads and indices_prefetched and hitpts.is_cuda
        main = torch.cuda.current_stream() if hitpts.is_cuda else None
        if fork:
            for st in self._heads_streams():
                st.wait_stream(main)
            self._cross_stream([point_inv_feat, hitpts], self._heads_streams())      # read by the branches after this function returned (deferred join)

        def branch(k, fn):
            if not fork:
                return fn()
            with torch.cuda.stream(self._head_streams[k]):
                out = fn()
            self._cross_stream(list(out) if isinstance(out, tuple) else [out], [main])
            return out

        # The direction head is ENQUEUED first: a handful of chip-wide launches on the current stream, against ~200 small launches for the two nets.
        # When the host is ahead of the GPU (the pipeline's normal state) the order is immaterial; when it is not (a busy host, a profiler) the current
        # stream would otherwise idle for the whole time the host spends enqueueing the nets (seen as 15 - 70 ms gaps in front of
        # interp_schedule_kernel in profiles/r05_stream_timeline.txt).  The side streams forked above: they do not wait for these kernels.
        direction = None
        if "direction" in pred_items and self.direction_first:
            if direction_mode != "standard_vector":
                raise AssertionError("Not implemented")   # same as the reference (:199,210)
            standard_vector = self._const(('sv', B, N, self.standard_vector.data_ptr(), self.standard_vector._version), lambda: self.standard_vector.repeat(B, N, 1))
            direction = self.decode_direction(None, so3_anchors, standard_vector, tokens_cl=point_equiv_cl, interp=interp)
        if "confidence" in pred_items:
            part_labels, confidences = branch(0, lambda: self.decode_confidence(point_inv_feat, hitpts))
            results["confidences"] = confidences
            results["part_labels"] = part_labels
        if "magnitude" in pred_items:
            results["magnitude"] = branch(1, lambda: self.decode_magnitude(point_inv_feat, hitpts))
        if "direction" in pred_items and not self.direction_first:
            if direction_mode != "standard_vector":
                raise AssertionError("Not implemented")
            direction = self.decode_direction(None, so3_anchors, self._const(('sv', B, N, self.standard_vector.data_ptr(), self.standard_vector._version), lambda: self.standard_vector.repeat(B, N, 1)), tokens_cl=point_equiv_cl, interp=interp)
        if direction is not None:
            results["direction"] = direction
        self.pending_join = None
        if fork:
            if self.defer_join:
                # a pipelined caller joins on ITS stage-2 stream: the current stream goes on to the next batch's encoder while the
                # tail of the confidence net (its 128 -> 11 008 -> 86 head, a chip-wide kernel) is still running
                self.pending_join = []
                for st in self._head_streams:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    self.pending_join.append(ev)
            else:
                for st in self._head_streams:
                    main.wait_stream(st)
