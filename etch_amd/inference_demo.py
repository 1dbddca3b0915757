"""MI355X counterpart of /root/reference/src/inference_demo.py: same CLI flags, same `load_model`,
`preprocess_scan`, `sample_points_from_mesh`, `predict_smpl`, same output files
(`<scan>_pred_smpl.obj`, `<scan>_output_smpl_info.npz` with body_pose, hand_pose, betas, global_orient, transl, joints).

Differences forced by the environment (no trimesh here): OBJ parsing and area-weighted surface sampling are
small numpy routines (the reference's sampling is unseeded, inference_demo.py:38; ours takes --seed), and the SMPL
body model comes from --body_model (a chumpy-free SMPL pickle) or --synthetic_body (seeded SMPL-shaped stand-in).
"""
import argparse
import json
import os

import numpy as np
import torch

from . import ops
from .models.fit_SMPL import Mesh, fit_smpl
from .models.models_pointcloud import GT_network_equiv


def load_model(args):
    """inference_demo.py:12-17."""
    model = GT_network_equiv(option=args).to(args.device)
    if args.model_path:
        model.load_state_dict(torch.load(args.model_path, map_location="cpu"))
    model.eval()
    return model


def load_obj(path):
    vs, fs = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                vs.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(t.split("/")[0]) - 1 for t in line.split()[1:]]
                for k in range(1, len(idx) - 1):
                    fs.append([idx[0], idx[k], idx[k + 1]])
    return Mesh(np.asarray(vs, np.float64), np.asarray(fs, np.int64))


def preprocess_scan(scan_path):
    """inference_demo.py:19-34: centre on the bounding-box mid-point."""
    scan_mesh = load_obj(scan_path)
    v = scan_mesh.vertices
    scan_center = (np.min(v, axis=0) + np.max(v, axis=0)) / 2.0
    centered = scan_mesh.copy()
    centered.vertices = v - scan_center
    return centered, scan_center


def sample_points_from_mesh(mesh, num_points=5000, seed=0):
    """inference_demo.py:36-39 (trimesh.sample.sample_surface): area-weighted faces + uniform barycentric coordinates."""
    rng = np.random.default_rng(seed)
    tri = mesh.vertices[mesh.faces]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    face = rng.choice(len(area), size=num_points, p=area / area.sum())
    r = rng.random((num_points, 2))
    flip = r.sum(1) > 1
    r[flip] = 1 - r[flip]
    t = tri[face]
    return t[:, 0] + r[:, :1] * (t[:, 1] - t[:, 0]) + r[:, 1:] * (t[:, 2] - t[:, 0])


def predict_smpl_batch(args, model, points_tensor, gender="neutral", **fit_kwargs):
    """inference_demo.py:41-66 for a batch (B,N,3) already on the device: stage 1 -> labels / inner points -> stage 2."""
    with torch.no_grad():
        results, _ = model(points_tensor, pred_items=["confidence", "direction", "magnitude"], direction_mode="standard_vector")
        labels = ops.argmax_rows(results["part_labels"])
        inner = ops.inner_points(points_tensor.contiguous(), results["direction"], results["magnitude"], float(args.scale_magnitude))
        return fit_smpl(args, inner, labels, results["confidences"], gender, **fit_kwargs)


def predict_smpl(args, model, points, gender="neutral"):
    """inference_demo.py:41-66."""
    points_tensor = torch.from_numpy(points).float().unsqueeze(0).to(args.device)
    meshes, _, _, info = predict_smpl_batch(args, model, points_tensor, gender)
    return meshes[0], info


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--scan_path", type=str, required=True, help="Path to input scan OBJ file")
    parser.add_argument("--gender", type=str, default="neutral", choices=["neutral", "male", "female"])
    parser.add_argument("--model_path", type=str, default="", help="Path to trained model")
    parser.add_argument("--markerset_path", default="datafolder/useful_data_4d-dress/superset_smpl.json", type=str)
    parser.add_argument("--output_folder", type=str, default="output")
    parser.add_argument("--num_point", type=int, default=5000)
    parser.add_argument("--scale_magnitude", type=int, default=10)
    parser.add_argument("--EPN_input_radius", type=float, default=0.4)
    parser.add_argument("--EPN_layer_num", type=int, default=2, choices=[1, 2, 3, 4],
                        help="encoder depth (models_pointcloud.py:34-48: feature widths 32 / 64 / 128 / 256); the release uses 2")
    parser.add_argument("--body_model", type=str, default="", help="chumpy-free SMPL .pkl (else the reference's gender paths)")
    parser.add_argument("--synthetic_body", action="store_true", help="use the seeded SMPL-shaped stand-in body model")
    parser.add_argument("--seed", type=int, default=0, help="surface-sampling seed (the reference is unseeded)")
    args = parser.parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("etch_amd needs an AMD GPU (gfx950); there is no CPU path")
    args.cuda = True
    args.device = torch.device("cuda")
    if os.path.exists(args.markerset_path):
        with open(args.markerset_path) as f:
            args.markerset = json.load(f)
    else:
        from . import constants
        args.markerset = constants.default_markerset()
    if args.synthetic_body:
        from .utils.body_model import SyntheticSMPL
        args.body_model = SyntheticSMPL(7)
    elif args.body_model:
        from .utils.body_model import load_smpl_pkl, load_smplx
        # an SMPL-X model file (SMPLX_*.npz / .pkl) selects the 55-joint, 20-coefficient fit; anything else is read as an SMPL pickle
        args.body_model = load_smplx(args.body_model) if "smplx" in os.path.basename(args.body_model).lower() else load_smpl_pkl(args.body_model)
    else:
        args.body_model = None
    os.makedirs(args.output_folder, exist_ok=True)
    model = load_model(args)
    centered_mesh, original_center = preprocess_scan(args.scan_path)
    points = sample_points_from_mesh(centered_mesh, args.num_point, args.seed)
    pred_smpl_mesh, smpl_info = predict_smpl(args, model, points, args.gender)
    final = pred_smpl_mesh.copy()
    final.vertices = pred_smpl_mesh.vertices + original_center
    scan_name = os.path.splitext(os.path.basename(args.scan_path))[0]
    final.export(os.path.join(args.output_folder, f"{scan_name}_pred_smpl.obj"))
    np.savez(os.path.join(args.output_folder, f"{scan_name}_output_smpl_info.npz"), body_pose=smpl_info[0][0, :21, :],
             hand_pose=smpl_info[0][0, 21:23, :], betas=smpl_info[1][0], global_orient=smpl_info[2][0], transl=smpl_info[3][0],
             joints=smpl_info[4][0])
    print(f"Predicted SMPL mesh and smpl info saved under {args.output_folder}")


if __name__ == "__main__":
    main()
