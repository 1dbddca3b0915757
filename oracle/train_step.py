"""TEST / MEASUREMENT INFRASTRUCTURE (CPU oracle): one training step of the reference's train.py:77-124 -- forward of GT_network_equiv in train() mode
(BatchNorm on batch statistics), the four losses of train.py:81-101 with their default weights, torch.autograd backward, torch.optim.Adam(lr 1e-4,
train.py:219) -- on the CPU restatement of oracle/stage1.py.  Used only by tests/ and by bench.py's `cpu_baseline` leg of `--train`; the product never imports it."""
import time

import torch
import torch.nn.functional as F

from . import stage1 as S1

MLPS = ((32, 32), (64, 64), (128, 128), (256, 256))


def losses(res, vec, conf, labels, scale_magnitude=10.0):
    """train.py:81-101 (direction_w = magnitude_w = part_label_w = confidence_w = 1); selected_indexs = arange (models_pointcloud.py:179)."""
    out = {"direction_loss": (1 - F.cosine_similarity(vec, res["direction"], dim=-1)).mean(),
           "magnitude_loss": F.mse_loss(torch.norm(vec, dim=-1, keepdim=True) * scale_magnitude, res["magnitude"]),
           "confidence_loss": F.mse_loss(res["confidences"], conf),
           "part_label_loss": F.cross_entropy(res["part_labels"].permute(0, 2, 1).contiguous(), labels)}
    return sum(out.values()), out


def train_step(sd, names, pts, vec, conf, labels, depth=2, lr=1e-4, num_markers=86):
    """sd: state dict (fp32 CPU tensors; `names` = the trainable parameters, made leaves here).  -> (loss parts, seconds forward, backward, Adam)."""
    S1.BN_TRAINING = True
    try:
        for k in names:
            sd[k] = sd[k].detach().clone().requires_grad_()
        B, N, _ = pts.shape
        t0 = time.perf_counter()
        xyz, feats = S1.encoder_forward(sd, pts, S1.build_layer_table(mlps=MLPS[:depth], strides=(2,) * depth))
        S_ = xyz.shape[-1]
        pef = S1.feat_propagation(pts.permute(0, 2, 1), xyz, feats.permute(0, 1, 3, 2).reshape(B, -1, S_)).reshape(B, N, -1, 60)
        p = pts.reshape(-1, 3).contiguous()
        inv = pef.mean(-1).reshape(B * N, -1).contiguous()
        o = torch.tensor([N * (i + 1) for i in range(B)], dtype=torch.int32)
        res = {}
        res["part_labels"], res["confidences"] = S1.pt_confidence(sd, "confidence_encoder.", p, inv, o, num_markers)
        res["magnitude"] = S1.pt_magnitude(sd, "magnitude_encoder.", p, inv, o)
        aw = S1.direction_anchor_weights(sd, pef)
        R, _, _ = S1.so3_mean(sd[f"encoder.backbone.{depth - 1}.blocks.1.intra_conv.conv.anchors"], aw)
        res["direction"] = R[:, :, 2].reshape(B, N, 3)
        loss, parts = losses(res, vec, conf, labels)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        torch.optim.Adam([sd[k] for k in names if sd[k].grad is not None], lr=lr).step()
        t3 = time.perf_counter()
        return {k: float(v.detach()) for k, v in parts.items()}, t1 - t0, t2 - t1, t3 - t2
    finally:
        S1.BN_TRAINING = False
