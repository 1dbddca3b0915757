"""CPU restatement (PyTorch fp32) of ETCH stage 1: GT_network_equiv.forward.  TEST INFRASTRUCTURE ONLY.

Functional style: every function takes a flat state dict `sd` (reference key names) and follows
the cited reference lines (paths relative to /root/reference).  Index ops come from
oracle/discrete_ops.c.  Pinned by tests/golden/* (generated from the reference's own Python by
oracle/ref_harness/gen_golden.py); see tests/test_oracle_vs_golden.py.

  build_layer_table     src/models/so3net.py:36-152
  encoder_forward       src/models/so3net.py:23-33, src/models/so3conv.py:7-16,19-183,
                        external/vgtk/vgtk/so3conv/functional.py:51-105,159-185,224-324,331-378,
                        external/vgtk/vgtk/so3conv/modules.py:20-39,92-153, vgtk/pc/sample.py:50-89
  feat_propagation      src/models/pointnet2_utils.py:4-74
  direction_head        src/models/direction_backbones.py:6-223, models_pointcloud.py:111-126
  so3_mean              src/models/so3conv.py:186-225
  pt_*                  src/models/pointtransformer_seg.py:8-268, src/models/pointops.py:10-178
  forward               src/models/models_pointcloud.py:146-221
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as O

NA = 60


# ----------------------------------------------------------------------------- layer table
def build_layer_table(input_radius=0.4, mlps=((32, 32), (64, 64)), strides=(2, 2), input_num=1024,
                      initial_radius_ratio=0.2, sampling_ratio=0.8, sampling_density=0.5, sigma_ratio=0.5):
    """so3net.py:36-152 with the defaults GT_network_equiv passes (models_pointcloud.py:31-49)."""
    strides = list(strides)
    if input_num > 1024:
        sampling_ratio /= input_num / 1024
        strides[0] = int(2 * (input_num / 1024))
    n_layer = len(mlps)
    mult = [1]
    for _ in range(n_layer):
        mult.append(mult[-1] * 2)
    num_centers = [int(input_num / m) for m in mult]
    radius_ratio = [initial_radius_ratio * m ** sampling_density for m in mult]
    radii = [r * input_radius for r in radius_ratio]
    sigma = [sigma_ratio * radii[0] ** 2]
    for i, s in enumerate(strides):
        sigma.append(sigma[i] * s)
    table, dim_in = [], 1
    for i, block in enumerate(mlps):
        bp = []
        for j, dim_out in enumerate(block):
            lazy = i != 0 or j != 0
            neighbor = int(sampling_ratio * num_centers[i] * radius_ratio[i] ** (1 / sampling_density))
            if i == 0 and j == 0:
                neighbor *= int(input_num / 1024)
            if j == 0:
                stride, nidx = strides[i], (i if i == 0 else i + 1)
                neighbor *= 2
            else:
                stride, nidx = 1, i + 1
            bp.append(dict(dim_in=dim_in, dim_out=dim_out, stride=stride, radius=radii[nidx], sigma=sigma[nidx],
                           n_neighbor=neighbor, lazy_sample=lazy))
            dim_in = dim_out
        table.append(bp)
    return table


# ----------------------------------------------------------------------------- encoder
def _inorm_lrelu(x):
    return F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.01)


def basic_so3conv(W, bias, x):
    """modules.py:33-39: x [b, c1, k, p, a] -> [b, c2, p, a]."""
    bs, npnt, na = x.shape[0], x.shape[3], x.shape[4]
    x = x.reshape(bs, W.shape[1], npnt * na)
    x = torch.matmul(W, x) + bias
    return x.view(bs, W.shape[0], npnt, na)


def inter_grouping(xyz, stride, radius, n_neighbor, lazy):
    """functional.py:176-185 + pc/sample.py:58-89.  xyz [b,3,p1] -> grouped_xyz [b,3,p2,nn], ball_idx, sample_idx, new_xyz."""
    b, _, p1 = xyz.shape
    n_sample = math.ceil(p1 / stride)
    if p1 == n_sample or lazy:
        sidx = torch.arange(n_sample).view(1, -1).expand(b, -1).int().contiguous()
    else:
        sidx = torch.from_numpy(O.furthest_point_sampling(xyz.numpy(), n_sample))
    new_xyz = torch.from_numpy(O.gather_points_forward(xyz.numpy(), sidx.numpy()))
    ball = torch.from_numpy(O.ball_query(new_xyz.numpy(), xyz.numpy(), float(radius), int(n_neighbor)))
    shadow = torch.cat((xyz, torch.ones(b, 3, 1) * 1e4), dim=2).contiguous()
    g = torch.from_numpy(O.gather_points_forward(shadow.numpy(), ball.view(b, -1).numpy())).view(b, 3, n_sample, n_neighbor)
    return g - new_xyz.unsqueeze(3), ball, sidx, new_xyz


def inter_weights(grouped_xyz, anchors, kernels, sigma):
    """functional.py:286-324: -> [b, p2, na, ks, nn]."""
    rk = torch.matmul(anchors, kernels.transpose(0, 1)).permute(1, 0, 2).contiguous()  # 3, na, ks
    t_rk = rk[None, :, None, :, :, None]
    t_g = grouped_xyz[..., None, None, :]
    d = torch.sum((t_g - t_rk) ** 2, dim=1)
    return F.relu(1.0 - d / sigma)


def inter_feat_grouping(ball, w, feats):
    """functional.py:61-67 on shadow-padded feats [b,c,q+1,a]."""
    b, p, nn = ball.shape
    _, c, q, a = feats.shape
    idx = ball.long().view(b, 1, p * nn, 1).expand(b, c, p * nn, a)
    nf = torch.gather(feats, 2, idx).view(b, c, p, nn, a)
    return torch.einsum("bcpna,bpakn->bckpa", nf, w).contiguous()


def separable_block(sd, pre, xyz, feats, cfg, chunk=256):
    """so3conv.py:171-183.  Returns new_xyz [b,3,p2], feats [b,c2,p2,60], sample_idx, ball_idx."""
    anchors = sd[pre + "inter_conv.conv.anchors"]
    kernels = sd[pre + "inter_conv.conv.kernels"]
    g, ball, sidx, new_xyz = inter_grouping(xyz, cfg["stride"], cfg["radius"], cfg["n_neighbor"], cfg["lazy_sample"])
    b, c, q, a = feats.shape
    featsp = torch.cat((feats, torch.zeros(b, c, 1, a)), dim=2).contiguous()
    p2 = g.shape[2]
    outs = []
    for s in range(0, p2, chunk):  # chunked only to bound memory; per-point results are independent
        w = inter_weights(g[:, :, s:s + chunk], anchors, kernels, cfg["sigma"])
        outs.append(inter_feat_grouping(ball[:, s:s + chunk], w, featsp))
    nf = torch.cat(outs, dim=3)  # [b, c, ks, p2, a]
    x = basic_so3conv(sd[pre + "inter_conv.conv.basic_conv.W"], sd[pre + "inter_conv.conv.basic_conv.bias"], nf)
    x = _inorm_lrelu(x)
    # intra (functional.py:331-378, modules.py:150-153)
    intra_idx = sd[pre + "intra_conv.conv.intra_idx"]
    nb, cin, nq, na = x.shape
    f1 = x.index_select(3, intra_idx.view(-1)).view(nb, cin, nq, na, intra_idx.shape[1])
    gf = f1.permute(0, 1, 4, 2, 3).contiguous()
    x = basic_so3conv(sd[pre + "intra_conv.conv.basic_conv.W"], sd[pre + "intra_conv.conv.basic_conv.bias"], gf)
    x = _inorm_lrelu(x)
    # skip (so3conv.py:178-182)
    skip = feats
    if cfg["stride"] > 1:
        ii = sidx.long().view(b, 1, -1, 1).expand(b, c, -1, a)
        skip = torch.gather(skip, 2, ii)
    skip = F.conv2d(skip, sd[pre + "skip_conv.weight"], sd[pre + "skip_conv.bias"])
    skip = _inorm_lrelu(skip)
    return new_xyz, x + skip, sidx, ball


def encoder_forward(sd, hitpts, table, prefix="encoder."):
    """so3net.py:23-33.  hitpts [B,N,3] -> xyz [B,3,P], feats [B,C,P,60]."""
    xyz = hitpts.permute(0, 2, 1).contiguous()
    feats = torch.ones(hitpts.shape[0], 1, hitpts.shape[1], NA)
    for bi, block in enumerate(table):
        for ci, cfg in enumerate(block):
            xyz, feats, _, _ = separable_block(sd, f"{prefix}backbone.{bi}.blocks.{ci}.", xyz, feats, cfg)
    return xyz, feats


# ----------------------------------------------------------------------------- propagation
def square_distance(src, dst):
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(src.shape[0], -1, 1)
    dist += torch.sum(dst ** 2, -1).view(dst.shape[0], 1, -1)
    return dist


def feat_propagation(xyz1, xyz2, points2, return_aux=False):
    """pointnet2_utils.py:45-74.  xyz1 [B,3,N], xyz2 [B,3,S], points2 [B,D,S] -> [B,N,D]."""
    xyz1 = xyz1.permute(0, 2, 1)
    xyz2 = xyz2.permute(0, 2, 1)
    points2 = points2.permute(0, 2, 1)
    B, N, _ = xyz1.shape
    dists = square_distance(xyz1, xyz2)
    dists, idx = dists.sort(dim=-1)
    dists, idx = dists[:, :, :3], idx[:, :, :3]
    dist_recip = 1.0 / (dists + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    bi = torch.arange(B).view(B, 1, 1).expand(B, N, 3)
    out = torch.sum(points2[bi, idx, :] * weight.view(B, N, 3, 1), dim=2)
    if return_aux:
        return out, idx, weight
    return out


# ----------------------------------------------------------------------------- direction head
def mhsa_layer(sd, pre, x, heads=8):
    """direction_backbones.py:132-194.  x [T,60,E]."""
    T, L, E = x.shape
    hs = E // heads
    k = F.linear(x, sd[pre + "key_transform.weight"])
    q = F.linear(x, sd[pre + "query_transform.weight"])
    v = F.linear(x, sd[pre + "value_transform.weight"])

    def split(o):
        return o.view(T, L, heads, hs).permute(2, 0, 1, 3).contiguous().view(T * heads, L, hs)

    k, q, v = split(k), split(q), split(v)
    logits = torch.bmm(q, k.permute(0, 2, 1)) / np.sqrt(hs)
    att = torch.bmm(F.softmax(logits, dim=-1), v)
    att = att.view(heads, T, L, hs).permute(1, 2, 0, 3).contiguous().view(T, L, E)
    return F.linear(att, sd[pre + "head_combine.weight"], sd[pre + "head_combine.bias"])


def direction_anchor_weights(sd, equiv_feat):
    """models_pointcloud.py:115-117.  equiv_feat [B,N,C,60] -> anc_w [B*N,60]."""
    B, N, C, na = equiv_feat.shape
    x = equiv_feat.permute(0, 1, 3, 2).reshape(-1, na, C).contiguous()
    y = mhsa_layer(sd, "direction_encoder.self_attention_layers.0.", x)
    x = x + y
    x = mhsa_layer(sd, "direction_encoder.self_attention_layers.1.", x)
    x = F.linear(x, sd["direction_predictor.net.0.weight"], sd["direction_predictor.net.0.bias"])
    x = F.relu(x)
    x = F.linear(x, sd["direction_predictor.net.2.weight"], sd["direction_predictor.net.2.bias"])
    w = F.conv1d(x.permute(0, 2, 1), sd["so3_reg.weight"], sd["so3_reg.bias"])
    return w.squeeze(1)


def so3_mean(anchors, weights):
    """so3conv.py:186-225 for Rs = anchors broadcast over the batch.  weights [T,60] -> R [T,3,3]."""
    Ce = torch.sum(weights[:, :, None, None] * anchors[None], dim=1)  # same reduction as the reference
    cu, cd, cv = torch.svd(Ce)
    cvT = cv.transpose(1, 2).contiguous()
    dets = torch.det(torch.matmul(cu, cvT))
    D = torch.zeros(Ce.shape[0], 3, 3)
    D[:, 0, 0] = 1
    D[:, 1, 1] = 1
    D[:, 2, 2] = dets
    return torch.einsum("bij,bjk,bkl->bil", cu, D, cvT), Ce, cd


# ----------------------------------------------------------------------------- point transformer
BN_TRAINING = False     # True: BatchNorm1d on batch statistics (model.train(), train.py:61) -- the running statistics are not touched here


def _bn(sd, pre, x):
    """BatchNorm1d on channel dim 1 of [n,c] or [n,c,l]: eval mode (running statistics) or, with BN_TRAINING, train mode (batch statistics)."""
    if BN_TRAINING:
        return F.batch_norm(x, None, None, sd[pre + "weight"], sd[pre + "bias"], True, 0.0, 1e-5)
    return F.batch_norm(x, sd[pre + "running_mean"], sd[pre + "running_var"], sd[pre + "weight"], sd[pre + "bias"], False, 0.0, 1e-5)


def _knn(k, p, new_p, o, new_o):
    i, d2 = O.knnquery(k, p.numpy(), new_p.numpy(), o.numpy(), new_o.numpy())
    return torch.from_numpy(i), torch.sqrt(torch.from_numpy(d2))


def _queryandgroup(ns, xyz, new_xyz, feat, o, new_o, use_xyz=True):
    idx, _ = _knn(ns, xyz, new_xyz, o, new_o)
    m, c = new_xyz.shape[0], feat.shape[1]
    gx = xyz[idx.view(-1).long(), :].view(m, ns, 3) - new_xyz.unsqueeze(1)
    gf = feat[idx.view(-1).long(), :].view(m, ns, c)
    return torch.cat((gx, gf), -1) if use_xyz else gf


def pt_layer(sd, pre, p, x, o, ns, share=8):
    """pointtransformer_seg.py:25-37."""
    xq = F.linear(x, sd[pre + "linear_q.weight"], sd[pre + "linear_q.bias"])
    xk = F.linear(x, sd[pre + "linear_k.weight"], sd[pre + "linear_k.bias"])
    xv = F.linear(x, sd[pre + "linear_v.weight"], sd[pre + "linear_v.bias"])
    xk = _queryandgroup(ns, p, p, xk, o, o, True)
    xv = _queryandgroup(ns, p, p, xv, o, o, False)
    pr, xk = xk[:, :, 0:3], xk[:, :, 3:]
    pr = F.linear(pr, sd[pre + "linear_p.0.weight"], sd[pre + "linear_p.0.bias"])
    pr = _bn(sd, pre + "linear_p.1.", pr.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()
    pr = F.relu(pr)
    pr = F.linear(pr, sd[pre + "linear_p.3.weight"], sd[pre + "linear_p.3.bias"])
    w = xk - xq.unsqueeze(1) + pr
    w = _bn(sd, pre + "linear_w.0.", w.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()
    w = F.relu(w)
    w = F.linear(w, sd[pre + "linear_w.2.weight"], sd[pre + "linear_w.2.bias"])
    w = _bn(sd, pre + "linear_w.3.", w.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()
    w = F.relu(w)
    w = F.linear(w, sd[pre + "linear_w.5.weight"], sd[pre + "linear_w.5.bias"])
    w = F.softmax(w, dim=1)
    n, nsample, c = xv.shape
    return ((xv + pr).view(n, nsample, share, c // share) * w.unsqueeze(2)).sum(1).view(n, c)


def pt_block(sd, pre, p, x, o, ns):
    """pointtransformer_seg.py:113-122."""
    idt = x
    x = F.relu(_bn(sd, pre + "bn1.", F.linear(x, sd[pre + "linear1.weight"])))
    x = F.relu(_bn(sd, pre + "bn2.", pt_layer(sd, pre + "transformer2.", p, x, o, ns)))
    x = _bn(sd, pre + "bn3.", F.linear(x, sd[pre + "linear3.weight"]))
    return F.relu(x + idt)


def pt_down(sd, pre, p, x, o, stride, ns):
    """pointtransformer_seg.py:52-68."""
    if stride != 1:
        ol = o.tolist()
        n_o, cnt = [ol[0] // stride], ol[0] // stride
        for i in range(1, len(ol)):
            cnt += (ol[i] - ol[i - 1]) // stride
            n_o.append(cnt)
        n_o = torch.tensor(n_o, dtype=torch.int32)
        idx = torch.from_numpy(O.furthestsampling(p.numpy(), o.numpy(), n_o.numpy()))
        n_p = p[idx.long(), :]
        g = _queryandgroup(ns, p, n_p, x, o, n_o, True)
        y = F.relu(_bn(sd, pre + "bn.", F.linear(g, sd[pre + "linear.weight"]).transpose(1, 2).contiguous()))
        y = F.max_pool1d(y, ns).squeeze(-1)
        return n_p, y, n_o
    return p, F.relu(_bn(sd, pre + "bn.", F.linear(x, sd[pre + "linear.weight"]))), o


def pt_interpolation(xyz, new_xyz, feat, o, new_o, k=3):
    """pointops.py:164-178 (weights from NON-squared distances)."""
    idx, dist = _knn(k, xyz, new_xyz, o, new_o)
    r = 1.0 / (dist + 1e-8)
    w = r / torch.sum(r, dim=1, keepdim=True)
    out = torch.zeros(new_xyz.shape[0], feat.shape[1])
    for i in range(k):
        out += feat[idx[:, i].long(), :] * w[:, i].unsqueeze(-1)
    return out


def pt_up(sd, pre, pxo1, pxo2=None):
    """pointtransformer_seg.py:81-98."""
    if pxo2 is None:
        _, x, o = pxo1
        ol = [0] + o.tolist()
        parts = []
        for i in range(len(ol) - 1):
            xb = x[ol[i]:ol[i + 1], :]
            cnt = ol[i + 1] - ol[i]
            g = F.relu(F.linear(xb.sum(0, True) / cnt, sd[pre + "linear2.0.weight"], sd[pre + "linear2.0.bias"]))
            parts.append(torch.cat((xb, g.repeat(cnt, 1)), 1))
        x = torch.cat(parts, 0)
        return F.relu(_bn(sd, pre + "linear1.1.", F.linear(x, sd[pre + "linear1.0.weight"], sd[pre + "linear1.0.bias"])))
    p1, x1, o1 = pxo1
    p2, x2, o2 = pxo2
    a = F.relu(_bn(sd, pre + "linear1.1.", F.linear(x1, sd[pre + "linear1.0.weight"], sd[pre + "linear1.0.bias"])))
    b = F.relu(_bn(sd, pre + "linear2.1.", F.linear(x2, sd[pre + "linear2.0.weight"], sd[pre + "linear2.0.bias"])))
    return a + pt_interpolation(p2, p1, b, o2, o1)


PT_BLOCKS = (2, 3, 4, 6, 3)
PT_STRIDE = (1, 4, 4, 4, 4)
PT_NS = (8, 16, 16, 16, 16)


def pt_unet(sd, pre, p0, x0, o0):
    """Shared encoder/decoder of both Point-Transformer nets (pointtransformer_seg.py:163-178,237-252)."""
    x = torch.cat((p0, x0), 1)
    levels = []
    p, o = p0, o0
    for li in range(5):
        e = f"{pre}enc{li + 1}."
        p, x, o = pt_down(sd, e + "0.", p, x, o, PT_STRIDE[li], PT_NS[li])
        for bi in range(1, PT_BLOCKS[li]):
            x = pt_block(sd, f"{e}{bi}.", p, x, o, PT_NS[li])
        levels.append([p, x, o])
    p5, x5, o5 = levels[4]
    x5 = pt_block(sd, pre + "dec5.1.", p5, pt_up(sd, pre + "dec5.0.", [p5, x5, o5]), o5, PT_NS[4])
    levels[4][1] = x5
    for li in (3, 2, 1, 0):
        pl, xl, ol = levels[li]
        pc, xc, oc = levels[li + 1]
        d = f"{pre}dec{li + 1}."
        xl = pt_block(sd, d + "1.", pl, pt_up(sd, d + "0.", [pl, xl, ol], [pc, xc, oc]), ol, PT_NS[li])
        levels[li][1] = xl
    return levels[0][1]


def pt_confidence(sd, pre, p0, x0, o0, k):
    """pointtransformer_seg.py:163-195."""
    B = len(o0)
    N = p0.shape[0] // B
    x1 = pt_unet(sd, pre, p0, x0, o0).reshape(B, N, -1).permute(0, 2, 1).contiguous()
    h = F.relu(_bn(sd, pre + "cls.1.", F.conv1d(x1, sd[pre + "cls.0.weight"], sd[pre + "cls.0.bias"])))
    logits = F.conv1d(h, sd[pre + "cls.3.weight"], sd[pre + "cls.3.bias"])
    sm = F.softmax(logits, dim=1)
    c = F.relu(F.conv1d(x1, sd[pre + "confi.0.weight"], sd[pre + "confi.0.bias"]))
    c = F.conv1d(c, sd[pre + "confi.2.weight"], sd[pre + "confi.2.bias"], groups=k)
    conf = (c.view(B, 1, k, N) * sm.unsqueeze(1)).sum(dim=2)
    return logits.permute(0, 2, 1).contiguous(), conf.permute(0, 2, 1).contiguous()


def pt_magnitude(sd, pre, p0, x0, o0):
    """pointtransformer_seg.py:237-260."""
    B = len(o0)
    N = p0.shape[0] // B
    x1 = pt_unet(sd, pre, p0, x0, o0)
    h = F.relu(_bn(sd, pre + "final_layer.1.", F.linear(x1, sd[pre + "final_layer.0.weight"], sd[pre + "final_layer.0.bias"])))
    return F.linear(h, sd[pre + "final_layer.3.weight"], sd[pre + "final_layer.3.bias"]).reshape(B, N, 1)


# ----------------------------------------------------------------------------- whole model
def forward(sd, hitpts, table, num_markers=86, return_aux=False):
    """models_pointcloud.py:146-221 with pred_items = all three, direction_mode='standard_vector'."""
    with torch.no_grad():
        B, N, _ = hitpts.shape
        xyz, feats = encoder_forward(sd, hitpts, table)
        S = xyz.shape[-1]
        equiv = feats.permute(0, 1, 3, 2).reshape(B, -1, S)
        pef = feat_propagation(hitpts.permute(0, 2, 1), xyz, equiv).reshape(B, N, -1, NA)
        inv = pef.mean(-1)
        p = hitpts.reshape(-1, 3).contiguous()
        x = inv.reshape(B * N, -1).contiguous()
        o = torch.tensor([N * (i + 1) for i in range(B)], dtype=torch.int32)
        labels, conf = pt_confidence(sd, "confidence_encoder.", p, x, o, num_markers)
        anc_w = direction_anchor_weights(sd, pef)
        # r.anchors of models_pointcloud.py:162 = the anchors buffer of the LAST conv the cloud went through: the intra conv of the last
        # separable block (so3conv.py:176-182 passes x.anchors on; vgtk modules.py:153 IntraSO3Conv returns self.anchors)
        last = max((k for k in sd if k.startswith("encoder.backbone.") and k.endswith(".intra_conv.conv.anchors")),
                   key=lambda k: tuple(int(t) for t in k.split(".") if t.isdigit()))
        anchors = sd[last]
        R, Ce, sv = so3_mean(anchors, anc_w)
        direction = R.reshape(B, N, 3, 3)[..., 2].contiguous()  # R @ [0,0,1]
        mag = pt_magnitude(sd, "magnitude_encoder.", p, x, o)
        res = {"part_labels": labels, "confidences": conf, "direction": direction, "magnitude": mag}
        if return_aux:
            res.update(anc_w=anc_w.view(B, N, NA), enc_xyz=xyz, enc_feats=feats, inv_feat=inv, sv=sv.view(B, N, 3))
        return res
