"""MI355X counterpart of /root/reference/src/models/fit_SMPL.py: `get_markers` and `fit_smpl` with the same
signatures and return values.  The Theseus / smplx machinery of the reference (autograd Jacobian through the full
6890-vertex LBS, dense Cholesky LM) is replaced by one persistent HIP kernel per batch (csrc/smpl_fit.hip).

Body model: the reference loads the licensed SMPL pickle by gender (fit_SMPL.py:92-101).  Here the model comes from
`args.body_model` (an etch_amd.utils.body_model.BodyModel, e.g. SyntheticSMPL or load_smpl_pkl(...)); if that
attribute is absent the reference's pickle paths are tried."""
import os

import numpy as np
import torch

from .. import ops
from ..utils.body_model import BodyModel, load_smpl_pkl

_GENDER_PATHS = {  # fit_SMPL.py:92-97
    "neutral": "datafolder/body_models/smpl/neutral/SMPL_NEUTRAL_10pc_rmchumpy.pkl",
    "female": "datafolder/body_models/smpl/female/SMPL_FEMALE_10pc.pkl",
    "male": "datafolder/body_models/smpl/male/SMPL_MALE_10pc.pkl",
}


class Mesh:
    """Minimal stand-in for the trimesh.Trimesh objects the reference returns (vertices, faces, copy, export to OBJ)."""

    def __init__(self, vertices, faces, process=False, maintain_order=True):
        self.vertices = np.asarray(vertices)
        self.faces = np.asarray(faces)

    def copy(self):
        return Mesh(self.vertices.copy(), self.faces.copy())

    def export(self, path):
        with open(path, "w") as f:
            for v in self.vertices:
                f.write("v %.8f %.8f %.8f\n" % (v[0], v[1], v[2]))
            for t in self.faces:
                f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


class _DeviceBody:
    """Device-resident tables of one (body model, marker set) pair, built once and cached on the BodyModel object."""

    def __init__(self, bm, marker_vids, device):
        d = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(device).contiguous()
        vids = np.asarray(marker_vids, np.int64)
        V = bm.num_verts
        self.nj, self.nb = bm.num_joints, bm.num_betas
        if (self.nj, self.nb) not in ((24, 10), (55, 20)):
            raise ValueError(f"the LM kernel is built for SMPL (24 joints, 10 betas) and an SMPL-X-sized model (55, 20); got ({self.nj}, {self.nb})")
        J0 = bm.J_regressor.astype(np.float64) @ bm.v_template.astype(np.float64)                       # (nj,3)
        Jd = np.einsum("jv,vcl->jcl", bm.J_regressor.astype(np.float64), bm.shapedirs.astype(np.float64))  # (nj,3,nb)
        P = bm.posedirs.reshape(9 * (self.nj - 1), V, 3)
        self.M = len(vids)
        self.V = V
        self.n_extra = len(bm.extra_vids)
        self.faces = bm.faces
        self.J0, self.Jd, self.parents = d(J0), d(Jd), d(bm.parents, torch.int32)
        # marker posedirs table for the LM kernel: (M, nj-1, 28) = per joint the 9 x 3 block (entry (e, a) at 3 e + a) padded to 28 floats
        Pm = np.transpose(P[:, vids, :], (1, 0, 2)).reshape(len(vids), self.nj - 1, 27)
        Pm = np.concatenate([Pm, np.zeros((len(vids), self.nj - 1, 1), Pm.dtype)], 2)
        self.lm_consts = [self.J0, self.Jd, self.parents, d(bm.v_template[vids]), d(bm.shapedirs[vids]), d(Pm), d(bm.lbs_weights[vids])]
        self.lbs_consts = [d(bm.v_template), d(bm.shapedirs), d(bm.posedirs), d(bm.lbs_weights), self.J0, self.Jd, self.parents,
                           d(bm.extra_vids, torch.int32)]


def _device_body(bm, marker_vids, device):
    key = (tuple(int(v) for v in marker_vids), str(device))
    cache = bm.__dict__.setdefault("_etch_device", {})
    if key not in cache:
        cache[key] = _DeviceBody(bm, marker_vids, device)
    return cache[key]


def _resolve_body_model(args, gender):
    bm = getattr(args, "body_model", None)
    if isinstance(bm, dict):
        bm = bm[gender]
    if bm is None:
        if gender not in _GENDER_PATHS:
            raise ValueError(f"Unexpected gender: {gender}")
        path = _GENDER_PATHS[gender]
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: provide args.body_model (etch_amd.utils.body_model) or the licensed SMPL pickle")
        bm = load_smpl_pkl(path)
    assert isinstance(bm, BodyModel)
    return bm


def get_markers(args, inner_points, part_labels, confidences):
    """fit_SMPL.py:17-62 -> pred_markers_position (B,M,3), valid_mask (B,M) bool."""
    M = len(args.markerset)
    markers, _, valid_b = ops.get_markers(inner_points.contiguous(), part_labels.contiguous(), confidences.contiguous(), M)
    return markers, valid_b


def fit_smpl_device(args, inner_points, part_labels, confidences, gender, steps_stage0=30, steps_stage1=50, lr_stage0=5e-1, lr_stage1=2e-1,
                    want_trace=False, markers_override=None):
    """Device-side part of fit_smpl: every kernel is enqueued on the current HIP stream, nothing is copied to the host.
    Returns a dict of device tensors (markers, valid, x, x_stage0, err_trace, verts, joints) + the face array.
    markers_override = (markers (B,M,3), valid_f (B,M) float, valid_b (B,M) bool): measurement aid of bench.py -- get_markers still runs
    on the network's output, but the fit consumes these markers instead (a seeded random network yields ~2 valid markers per scan and a
    fit that freezes early; see bench.py `well_posed_fit`)."""
    M = len(args.markerset)
    vids = list(args.markerset.values())
    bm = _resolve_body_model(args, gender)
    db = _device_body(bm, vids, inner_points.device)
    markers, valid_f, valid_b = ops.get_markers(inner_points.contiguous(), part_labels.contiguous(), confidences.contiguous(), M)
    if markers_override is not None:
        markers, valid_f, valid_b = markers_override
    # stage 0: damping 0.01 (fit_SMPL.py:200); stage 1: Theseus default damping 1e-3 (:249)
    x, x0, trace = ops.smpl_lm_fit(db.lm_consts, markers, valid_f, steps_stage0, lr_stage0, 0.01, steps_stage1, lr_stage1, 1e-3, want_trace,
                                   nj=db.nj, nb=db.nb)
    verts, joints = ops.smpl_lbs(db.lbs_consts, x, db.V, db.n_extra, nj=db.nj, nb=db.nb)
    # per-scan status word: bit 0 = NaN markers (conf**20 underflow, reproduced from the reference) -> NaN fit; bit 1 = no marker
    status = ops.marker_status(markers, valid_f)
    return dict(markers=markers, valid=valid_b, x=x, x_stage0=x0, err_trace=trace, verts=verts, joints=joints, faces=db.faces, status=status,
                nj=db.nj, nb=db.nb)


def fit_smpl_stage_host(dev, pinned=None):
    """Enqueue the device -> host copies of one fit on the CURRENT stream into pinned buffers (no host wait): a pipelined
    caller records an event after this and later finalizes without touching the stream again.  `pinned` = buffers of an
    earlier call to reuse (same shapes)."""
    host = {}
    for k in ("x", "verts", "joints"):
        t = dev[k].detach()
        buf = None if pinned is None else pinned.get(k)
        if buf is None or buf.shape != t.shape:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t, non_blocking=True)
        host[k] = buf
    dev["host"] = host
    return dev


def fit_smpl_finalize(dev):
    """Host-side part: the fit in the reference's return format (fit_SMPL.py:261-269).  Copies to the host here unless
    fit_smpl_stage_host already enqueued them (the caller has then waited for its event)."""
    B = dev["x"].shape[0]
    if "host" in dev:
        xn, vn, jn = (dev["host"][k].numpy().copy() for k in ("x", "verts", "joints"))
    else:
        xn, vn, jn = (dev[k].detach().cpu().numpy() for k in ("x", "verts", "joints"))
    meshes = [Mesh(vn[b], dev["faces"], process=False, maintain_order=True) for b in range(B)]
    npose, nb = 3 * (dev.get("nj", 24) - 1), dev.get("nb", 10)
    info = [xn[:, :npose].reshape(B, -1, 3), xn[:, npose:npose + nb].copy(), xn[:, npose + nb:npose + nb + 3].copy(),
            xn[:, npose + nb + 3:npose + nb + 6].copy(), jn]
    return meshes, dev["markers"], dev["valid"], info


def fit_smpl(args, inner_points, part_labels, confidences, gender, steps_stage0=30, steps_stage1=50, lr_stage0=5e-1, lr_stage1=2e-1,
             return_trace=False, markers_override=None):
    """fit_SMPL.py:68-269.  Returns (list of meshes, pred_markers_position (B,M,3), valid_mask (B,M) bool,
    [pose (B,23,3), shape (B,10), global_orient (B,3), translation (B,3), joints (B,45,3)] as numpy)."""
    dev = fit_smpl_device(args, inner_points, part_labels, confidences, gender, steps_stage0, steps_stage1, lr_stage0, lr_stage1, return_trace,
                          markers_override=markers_override)
    out = fit_smpl_finalize(dev)
    if return_trace:
        return out + (dev,)
    return out
