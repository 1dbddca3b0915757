"""Pinning the parts of the path this repository can only restate (the CUDA index kernels and the Theseus / smplx fit).

    python -m etch_amd.selfcheck dump.npz [--smpl-pkl SMPL_NEUTRAL.pkl]

`dump.npz` is written by the ~30-line script of INTEGRATION.md section 7 on ANY CUDA box inside the reference tree (it imports only the
reference's own modules: epn_grouping, pointops_cuda, models.fit_SMPL).  This module re-runs the same seeded inputs through libetch_hip.so on the
MI355X and compares: index results bit for bit, the fitted SMPL parameters at 1e-4.  Exit status 0 = every section present in the dump matches.

Sections (all optional; a missing key skips its section):
  vgtk FPS         fps_xyz (b,3,n) f32, fps_m -> fps_idx (b,m) i32                      grouping_cuda_kernel.cu:352-466
  ball query       bq_new_xyz (b,3,m), bq_xyz (b,3,n), bq_radius, bq_nsample -> bq_idx    grouping_cuda_kernel.cu:68-113
  pointops FPS     pfps_xyz (n,3), pfps_offset (b), pfps_new_offset (b) -> pfps_idx       sampling_cuda_kernel.cu:15-129
  pointops kNN     knn_xyz (n,3), knn_new_xyz (m,3), knn_offset, knn_new_offset, knn_nsample -> knn_idx (m,k), knn_dist (m,k) = sqrt(d2)
  marker fit       fit_inner (B,K,3), fit_labels (B,K) i64, fit_conf (B,K,1), fit_gender -> fit_pose (B,23,3), fit_betas (B,10), fit_orient, fit_transl,
                   fit_joints (B,45,3), fit_markers (B,86,3), fit_valid (B,86)              fit_SMPL.py:68-269 (needs --smpl-pkl: the licensed model)
"""
import argparse
import json
import sys
import types

import numpy as np
import torch


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def check(dump, smpl_pkl=None, markerset=None):
    from . import constants as K
    from . import ops
    report = {}

    def exact(name, got, want):
        got, want = got.cpu().numpy(), np.asarray(want)
        n_bad = int((got != want).sum())
        report[name] = {"match": n_bad == 0, "elements": int(want.size), "differing": n_bad}

    if "fps_idx" in dump:
        exact("vgtk_fps", ops.furthest_point_sampling(_dev(dump["fps_xyz"], torch.float32), int(dump["fps_m"])), dump["fps_idx"])
    if "bq_idx" in dump:
        exact("ball_query", ops.ball_query(_dev(dump["bq_new_xyz"], torch.float32), _dev(dump["bq_xyz"], torch.float32), float(dump["bq_radius"]),
                                           int(dump["bq_nsample"])), dump["bq_idx"])
    if "pfps_idx" in dump:
        exact("pointops_fps", ops.furthestsampling(_dev(dump["pfps_xyz"], torch.float32), _dev(dump["pfps_offset"], torch.int32),
                                                   _dev(dump["pfps_new_offset"], torch.int32)), dump["pfps_idx"])
    if "knn_idx" in dump:
        idx, dist = ops.knnquery(int(dump["knn_nsample"]), _dev(dump["knn_xyz"], torch.float32), _dev(dump["knn_new_xyz"], torch.float32),
                                 _dev(dump["knn_offset"], torch.int32), _dev(dump["knn_new_offset"], torch.int32))
        exact("pointops_knn_idx", idx, dump["knn_idx"])
        d = np.abs(dist.cpu().numpy() - np.asarray(dump["knn_dist"]))
        report["pointops_knn_dist"] = {"match": bool(d.max() <= 1e-6 * max(1.0, float(np.abs(dump["knn_dist"]).max()))), "max_abs_diff": float(d.max())}
    if "fit_pose" in dump:
        if smpl_pkl is None:
            report["marker_fit"] = {"match": None, "skipped": "needs --smpl-pkl (the licensed SMPL model the reference loaded)"}
        else:
            from .models.fit_SMPL import fit_smpl
            from .utils.body_model import load_smpl_pkl
            args = types.SimpleNamespace(markerset=markerset or K.default_markerset(), device=torch.device("cuda"), body_model=load_smpl_pkl(smpl_pkl))
            gender = str(dump["fit_gender"]) if "fit_gender" in dump else "neutral"
            _, markers, valid, info = fit_smpl(args, _dev(dump["fit_inner"], torch.float32), _dev(dump["fit_labels"], torch.int64),
                                               _dev(dump["fit_conf"], torch.float32), gender)
            dev = {"markers": float(np.nanmax(np.abs(markers.cpu().numpy() - dump["fit_markers"]))) if "fit_markers" in dump else None,
                   "valid_equal": bool((valid.cpu().numpy() == np.asarray(dump["fit_valid"]).astype(bool)).all()) if "fit_valid" in dump else None}
            for k, got in zip(("fit_pose", "fit_betas", "fit_orient", "fit_transl", "fit_joints"), info):
                if k in dump:
                    dev[k] = float(np.abs(got - np.asarray(dump[k])).max())
            worst = max(v for k, v in dev.items() if isinstance(v, float))
            report["marker_fit"] = {"match": bool(worst < 1e-4 and dev["valid_equal"] is not False), "max_abs_deviation": dev, "tolerance": 1e-4}
    return report


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("dump")
    ap.add_argument("--smpl-pkl", default=None, help="chumpy-free SMPL pickle of the gender the dump was made with")
    ap.add_argument("--markerset", default=None, help="superset_smpl.json (default: the 86-marker table shipped with etch_amd)")
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        sys.exit("etch_amd.selfcheck needs an MI355X: the product path has no CPU fallback")
    dump = dict(np.load(a.dump, allow_pickle=False))
    ms = json.load(open(a.markerset)) if a.markerset else None
    rep = check(dump, a.smpl_pkl, ms)
    print(json.dumps(rep, indent=1))
    bad = [k for k, v in rep.items() if v.get("match") is False]
    if bad:
        sys.exit(f"MISMATCH in: {', '.join(bad)}")
    if not rep:
        sys.exit("the dump holds none of the known sections")


if __name__ == "__main__":
    main()
