"""Batch evaluation harness around the hot path -- MI355X counterpart of the inference part of
/root/reference/src/eval.py:
  * per-gender batching of fit_smpl (eval.py:186-209): one call when the batch has a single gender, otherwise per gender
    group (the reference falls back to per-sample calls; grouping by gender gives the same per-sample results because
    every LM problem is independent, and keeps the batches large),
  * V2V = mean_v || v_gt - v_pred ||_2 over the 6 890 vertices (eval.py:235-237),
  * MPJPE over the first 22 joints (scripts/experiment_scripts/compute_mpjpe_error.py:23-24),
  * the per-sample output_smpl_info npz (eval.py:240-247) and v2v_score.txt lines (eval.py:254-256).
  * `EvalDataset` reads the reference's on-disk layout (README.md:74-90; src/data_utils/GT_dataloader_mixed.py:110-161,259-274):
    scans `<scan_dir>/<id>/<id>.obj`, body models `<smpl_dir>/<id>/{info_<id>.npz, mesh_smpl_<id>.obj}`, per-id `<infopoints_dir>/<id>.npz`
    and the pickled list of activated ids -- the inference-side fields only (points sampled on the scan, gender, ground-truth SMPL
    vertices / joints); the training targets of that loader (vectors, geodesic confidences, labels: cKDTree / potpourri3d / trimesh
    proximity queries) and eval.py's visualisation exports are out of scope.
  * `evaluate_dataset` is eval.py's loop (:76-265): batches, per-sample outputs, v2v_score.txt with its summary block."""
import os

import numpy as np
import torch

from . import ops
from .models.fit_SMPL import fit_smpl


class EvalDataset:
    """Inference-side view of the reference's GTDataset (GT_dataloader_mixed.py:110-161).  Items:
    {id, hitpts (num_point,3) float32, gender "female" | "male", gt_vertices (V,3), gt_joints (J,3) or None}."""

    gender_dict = {0: "female", 1: "male"}                # GT_dataloader_mixed.py:133

    def __init__(self, scan_dir, smpl_dir, infopoints_dir, activated_ids_path=None, num_point=5000, seed=0):
        import pickle
        self.scan_dir, self.smpl_dir, self.infopoints_dir = scan_dir, smpl_dir, infopoints_dir
        activated = None
        if activated_ids_path:
            with open(activated_ids_path, "rb") as f:
                activated = set(pickle.load(f))
        self.id_list = sorted(i for i in os.listdir(scan_dir)
                              if os.path.isdir(os.path.join(scan_dir, i)) and os.path.isdir(os.path.join(smpl_dir, i))
                              and os.path.isfile(os.path.join(infopoints_dir, f"{i}.npz")) and (activated is None or i in activated))
        self.num_point, self.seed = num_point, seed

    def __len__(self):
        return len(self.id_list)

    def __getitem__(self, index):
        from .inference_demo import load_obj, sample_points_from_mesh
        i = self.id_list[index]
        scan = load_obj(os.path.join(self.scan_dir, i, f"{i}.obj"))
        smpl = load_obj(os.path.join(self.smpl_dir, i, f"mesh_smpl_{i}.obj"))
        info = np.load(os.path.join(self.smpl_dir, i, f"info_{i}.npz"))
        # the reference samples with trimesh.sample.sample_surface(scan, num_point, seed = self.seed + 15) (:201-203); same
        # distribution (area-weighted faces, uniform barycentric coordinates) from our seeded sampler
        pts = sample_points_from_mesh(scan, self.num_point, seed=self.seed + 15).astype(np.float32)
        return {"id": i, "hitpts": pts, "gender": self.gender_dict[int(np.asarray(info["gender"]).item())],
                "gt_vertices": np.asarray(smpl.vertices, np.float32), "gt_joints": np.asarray(info["joints"], np.float32) if "joints" in info.files else None}


def evaluate_dataset(args, model, dataset, batch_size=8, output_folder=None):
    """eval.py:76-265: the whole dataset in batches through `evaluate_batch`; v2v_score.txt ends with the reference's summary block.
    Returns (records, average v2v)."""
    output_folder = output_folder or args.output_folder
    os.makedirs(output_folder, exist_ok=True)
    score = os.path.join(output_folder, "v2v_score.txt")
    if os.path.exists(score):
        os.remove(score)                                   # eval.py:82-84
    recs = []
    for s in range(0, len(dataset), batch_size):
        items = [dataset[k] for k in range(s, min(s + batch_size, len(dataset)))]
        hitpts = torch.from_numpy(np.stack([it["hitpts"] for it in items])).to(args.device)
        gtj = [it["gt_joints"] for it in items]
        recs += evaluate_batch(args, model, hitpts, [it["gender"] for it in items], ids=[it["id"] for it in items],
                               gt_vertices=[it["gt_vertices"] for it in items], gt_joints=None if any(j is None for j in gtj) else gtj,
                               output_folder=output_folder)
    total = float(sum(r["v2v"] for r in recs))
    n = len(recs)
    with open(score, "a") as f:                            # eval.py:258-265
        f.write("==========\n")
        f.write(f"average v2v: {total / max(n, 1)}\n")
        f.write(f"total v2v: {total}\n")
        f.write(f"sample num: {n}\n")
    return recs, total / max(n, 1)


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
