"""MI355X counterpart of /root/reference/src/models/fit_SMPL_Adam.py: the first-order (Adam) SMPL marker fit with the same
`get_markers` / `fit_smpl(args, inner_points, part_labels, confidences, steps_stage0=400, steps_stage1=800, lr=1e-2)` surface and the
same return value (list of meshes, pred_markers_position, valid_mask).  The reference differentiates the full-mesh SMPL forward with
autograd 1 200 times per batch; here one persistent HIP kernel per batch runs the whole schedule on chip, with the gradient
(2 / n) J^T r taken from the marker-restricted analytic linearisation of csrc/smpl_fit.hip (etch_smpl_adam_fit).

Divergence to know (csrc/smpl_fit.hip, smpl_adam_fit_kernel): the reference's mse_loss is one mean over the whole batch, so one scan with a
NaN marker makes every scan's result NaN; here a NaN stays in its own scan (flagged by ops.marker_status) and the other scans are fitted.
The returned meshes come from the parameters of the last forward pass (before the final optimizer step), as in the reference (:221-225),
also when steps_stage1 == 0.

Body model: the reference hard-codes the neutral SMPL pickle (fit_SMPL_Adam.py:92-94); here `args.body_model` (or its "neutral" entry)
is used, falling back to that pickle path."""
import numpy as np
import torch

from .. import ops
from .fit_SMPL import Mesh, _device_body, _resolve_body_model, get_markers  # noqa: F401  (get_markers: fit_SMPL_Adam.py:17-62, identical)


def fit_smpl(args, inner_points, part_labels, confidences, steps_stage0=400, steps_stage1=800, lr=1e-2, return_aux=False):
    """fit_SMPL_Adam.py:68-225."""
    M = len(args.markerset)
    bm = _resolve_body_model(args, "neutral")
    db = _device_body(bm, list(args.markerset.values()), inner_points.device)
    markers, valid_f, valid_b = ops.get_markers(inner_points.contiguous(), part_labels.contiguous(), confidences.contiguous(), M)
    x, x_last, trace = ops.smpl_adam_fit(db.lm_consts, markers, valid_f, steps_stage0, steps_stage1, lr, want_trace=return_aux, nj=db.nj, nb=db.nb)
    verts, joints = ops.smpl_lbs(db.lbs_consts, x_last, db.V, db.n_extra, nj=db.nj, nb=db.nb)      # the last forward of the loop (:221-225)
    vn = verts.detach().cpu().numpy()
    meshes = [Mesh(vn[b], db.faces, process=False, maintain_order=True) for b in range(vn.shape[0])]
    if return_aux:
        return meshes, markers, valid_b, dict(x=x, x_last=x_last, loss_trace=trace, verts=verts, joints=joints)
    return meshes, markers, valid_b
