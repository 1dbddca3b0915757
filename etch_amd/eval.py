"""Batch evaluation harness around the hot path -- MI355X counterpart of the inference part of
/root/reference/src/eval.py:
  * per-gender batching of fit_smpl (eval.py:186-209): one call when the batch has a single gender, otherwise per gender
    group (the reference falls back to per-sample calls; grouping by gender gives the same per-sample results because
    every LM problem is independent, and keeps the batches large),
  * V2V = mean_v || v_gt - v_pred ||_2 over the 6 890 vertices (eval.py:235-237),
  * MPJPE over the first 22 joints (scripts/experiment_scripts/compute_mpjpe_error.py:23-24),
  * the per-sample output_smpl_info npz (eval.py:240-247) and v2v_score.txt lines (eval.py:254-256).
Dataset readers / visualisation exports of eval.py are out of scope (trimesh / matplotlib side effects)."""
import os

import numpy as np
import torch

from . import ops
from .models.fit_SMPL import fit_smpl


def v2v(gt_vertices, pred_vertices):
    """eval.py:237."""
    return float(np.mean(np.linalg.norm(np.asarray(gt_vertices) - np.asarray(pred_vertices), axis=1)))


def mpjpe(gt_joints, pred_joints, joints_num_considered=22):
    """compute_mpjpe_error.py:23-24."""
    return float(np.linalg.norm(np.asarray(pred_joints)[:joints_num_considered] - np.asarray(gt_joints)[:joints_num_considered], axis=-1).mean())


def fit_smpl_by_gender(args, pred_inner_points, pred_part_labels, pred_confidences, gender_list, **fit_kwargs):
    """eval.py:186-209 with the same return layout as fit_smpl for the whole batch."""
    B = pred_inner_points.shape[0]
    assert len(gender_list) == B
    if len(set(gender_list)) == 1:
        return fit_smpl(args, pred_inner_points, pred_part_labels, pred_confidences, gender_list[0], **fit_kwargs)
    meshes = [None] * B
    markers = torch.zeros((B, len(args.markerset), 3), dtype=torch.float32, device=pred_inner_points.device)
    valid = torch.zeros((B, len(args.markerset)), dtype=torch.bool, device=pred_inner_points.device)
    info = None
    for g in sorted(set(gender_list)):
        sel = [i for i, x in enumerate(gender_list) if x == g]
        idx = torch.tensor(sel, device=pred_inner_points.device)
        m, mk, vm, inf = fit_smpl(args, pred_inner_points[idx].contiguous(), pred_part_labels[idx].contiguous(),
                                  pred_confidences[idx].contiguous(), g, **fit_kwargs)
        if info is None:
            info = [np.zeros((B,) + a.shape[1:], a.dtype) for a in inf]
        for k, i in enumerate(sel):
            meshes[i] = m[k]
            for a, src in zip(info, inf):
                a[i] = src[k]
        markers[idx] = mk
        valid[idx] = vm
    return meshes, markers, valid, info


def evaluate_batch(args, model, hitpts, gender_list, ids=None, gt_vertices=None, gt_joints=None, output_folder=None):
    """One eval.py iteration (eval.py:87-99,181-256) on a batch (B,K,3) already on the device.
    Returns a list of per-sample dicts {id, v2v, mpjpe, valid_full} (metrics only when ground truth is given)."""
    B = hitpts.shape[0]
    ids = list(ids) if ids is not None else [str(i) for i in range(B)]
    with torch.no_grad():
        results, _ = model(hitpts, pred_items=["confidence", "direction", "magnitude"], direction_mode="standard_vector")
        labels = ops.argmax_rows(results["part_labels"])
        inner = ops.inner_points(hitpts.contiguous(), results["direction"], results["magnitude"], float(args.scale_magnitude))
        meshes, markers, valid, info = fit_smpl_by_gender(args, inner, labels, results["confidences"], gender_list)
    out = []
    vsum = valid.sum(1).tolist()
    for j in range(B):
        rec = {"id": ids[j], "valid_full": int(vsum[j]) == valid.shape[1]}
        if gt_vertices is not None:
            rec["v2v"] = v2v(gt_vertices[j], meshes[j].vertices)
        if gt_joints is not None:
            rec["mpjpe"] = mpjpe(gt_joints[j], info[4][j])
        if output_folder is not None:
            d = os.path.join(output_folder, f"{ids[j]}")
            os.makedirs(d, exist_ok=True)
            meshes[j].export(os.path.join(d, f"forwarded_smpl_mesh_on_pred_{ids[j]}.obj"))
            np.savez(os.path.join(d, f"output_smpl_info_{ids[j]}.npz"), body_pose=info[0][j][:21, :], hand_pose=info[0][j][21:23, :],
                     betas=info[1][j], global_orient=info[2][j], transl=info[3][j], joints=info[4][j])
            if "v2v" in rec:
                with open(os.path.join(output_folder, "v2v_score.txt"), "a") as f:
                    f.write(f"{ids[j]}: {rec['v2v']}{'' if rec['valid_full'] else '  attention, the valid mask is not full'}\n")
        out.append(rec)
    return out
