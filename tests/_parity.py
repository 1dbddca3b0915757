"""Shared comparison helpers of the GPU parity tests."""
import numpy as np
import torch

RTOL = 1e-4      # north_star: within 1e-4 relative fp32 tolerance of the reference


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def direction_within_conditioning(direction, anc_w_gpu, anc_w_ref, direction_ref, anchors, min_tight=0.3):
    """direction = polar projection of Ce = sum_a w_a R_a (so3conv.py:186-225).  With random weights Ce is nearly singular
    (SURVEY H3), so the admissible deviation is the conditioning of the projection times the deviation of anc_w:
        |dR| <~ 2 |dCe|_F / gap,  dCe = sum_a (w_gpu - w_ref)_a R_a,  gap = min_{i<j} (s_i + s_j), s = (sv0, sv1, det * sv2).
    All arguments are numpy, flattened over points; returns the fraction of points whose bound is tight (< 0.05)."""
    from oracle import stage1 as S
    aw = torch.from_numpy(np.ascontiguousarray(anc_w_ref)).view(-1, 60)
    _, Ce, sv = S.so3_mean(torch.from_numpy(anchors), aw)
    det = torch.det(Ce).sign()
    s = torch.stack([sv[:, 0], sv[:, 1], det * sv[:, 2]], 1).double()
    gap = torch.stack([s[:, 0] + s[:, 1], s[:, 0] + s[:, 2], s[:, 1] + s[:, 2]], 1).min(1).values.clamp_min(1e-12).numpy()
    dw = anc_w_gpu.reshape(-1, 60).astype(np.float64) - anc_w_ref.reshape(-1, 60)
    dCe = np.linalg.norm(np.einsum("ta,aij->tij", dw, anchors.astype(np.float64)).reshape(-1, 9), axis=1)
    bound = 4.0 * dCe / gap + 1e-5
    err = np.abs(direction - direction_ref).reshape(-1, 3).max(1)
    tight = bound < 0.05
    assert tight.mean() > min_tight, tight.mean()
    assert (err[tight] <= bound[tight]).all(), float((err[tight] - bound[tight]).max())
    return float(tight.mean())


def check_stage1_vs_fixture(res, anc_w, g, anchors, scans=None, tol=RTOL):
    """GPU stage-1 outputs (dict of device tensors + the direction head's anchor weights) against a row-subsampled fixture written
    by gen_golden._save_model_run (reference Python).  `scans` = indices into the GPU batch that the fixture's scans correspond to."""
    rows = g["rows"]
    sel = slice(None) if scans is None else list(scans)
    out = {}
    for k, sub in (("part_labels", True), ("confidences", False), ("magnitude", False)):
        got = res[k][sel].cpu().numpy()
        got = got[:, rows] if sub else got
        assert got.shape == g[k].shape and np.isfinite(got).all(), k
        out[k] = rel_err(got, g[k])
        assert out[k] < tol, (k, out[k])
    labels = res["part_labels"][sel].argmax(-1).cpu().numpy()
    out["label_agreement"] = float((labels == g["labels"]).mean())
    assert out["label_agreement"] > 0.999, out["label_agreement"]      # an argmax may flip only where two logits tie within the tolerance
    aw = anc_w[sel].cpu().numpy()[:, rows]
    out["anc_w"] = rel_err(aw, g["anc_w"])
    assert out["anc_w"] < tol, out["anc_w"]
    d = res["direction"][sel].cpu().numpy()
    assert np.abs(np.linalg.norm(d, axis=-1) - 1).max() < 1e-4
    out["direction_tight_frac"] = direction_within_conditioning(d[:, rows], aw, g["anc_w"], g["direction"][:, rows], anchors)
    return out
