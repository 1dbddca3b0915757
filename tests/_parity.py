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
    if "anc_w_fp64" in g.files:
        # conditioning yardstick stored with the fixture: the reference's Python run in fp64 and, per point, how far the reference's
        # OWN fp32 run lands from it (a few points carry saturated attention rows that amplify fp32 rounding).  The GPU result
        # must be within tol of the fp32 reference on (nearly) as many points as the fp32 reference is within tol of the truth, and as
        # close to the fp64 truth as the reference's fp32 run is (x2 for a different summation order) everywhere.
        scale = float(g["anc_w_scale"])
        dev = np.abs(aw.astype(np.float64) - g["anc_w_fp64"]).max(-1) / scale
        ref = g["ref_fp32_dev"].astype(np.float64)
        per_point = np.abs(aw.astype(np.float64) - g["anc_w"]).max(-1) / np.abs(g["anc_w"]).max()
        out["anc_w_vs_fp64"] = {"gpu_max": float(dev.max()), "ref_fp32_max": float(ref.max()), "gpu_q999": float(np.quantile(dev, 0.999)),
                                "ref_fp32_q999": float(np.quantile(ref, 0.999))}
        out["anc_w_vs_fp64"]["points_within_tol_of_fp32_ref"] = float((per_point < tol).mean())
        assert (per_point < tol).mean() >= min(0.999, (ref < tol).mean() - 0.002), float((per_point < tol).mean())
        assert dev.max() <= max(2.0 * ref.max(), tol), out["anc_w_vs_fp64"]
        assert np.quantile(dev, 0.999) <= max(2.0 * np.quantile(ref, 0.999), tol), out["anc_w_vs_fp64"]
    else:
        assert out["anc_w"] < tol, out["anc_w"]
    d = res["direction"][sel].cpu().numpy()
    assert np.abs(np.linalg.norm(d, axis=-1) - 1).max() < 1e-4
    out["direction_tight_frac"] = direction_within_conditioning(d[:, rows], aw, g["anc_w"], g["direction"][:, rows], anchors)
    return out


def oracle_trace(trace, it0=30, it1=50):
    """Per-iteration error trace of oracle.stage2.fit_smpl as a (B, it0 + it1 + 2) array.  The oracle's LM loop ends as soon as
    EVERY sample of the batch has converged; the kernel keeps reporting the frozen error for the remaining iterations."""
    out = []
    for tr, n in zip(trace, (it0 + 1, it1 + 1)):
        t = torch.stack(tr, 1)
        if t.shape[1] < n:
            t = torch.cat([t, t[:, -1:].expand(-1, n - t.shape[1])], 1)
        out.append(t)
    return torch.cat(out, 1).numpy()
