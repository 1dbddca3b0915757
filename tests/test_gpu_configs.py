"""GPU parity at the sizes BASELINE.json's configs name, against fixtures emitted by the REFERENCE's own Python
(oracle/ref_harness/gen_golden.py: scan4d, model5k, model20k, padding_fp64) and, for stage 2, against the oracle run live.

  configs[0]  the bundled 4D-Dress scan, 5 000 surface points                 test_config0_bundled_scan_*
  configs[1]  batch 32 x 5 000 synthetic scans, equivariant forward           test_config1_2_batch32_* (stage-1 part)
  configs[2]  batch 32 x 5 000, forward + (30+50)-iteration LM fit            test_config1_2_batch32_* (stage-2 part)
  configs[4]  dense 20 000-point scans (geometry; SMPL-X fit: test_gpu_stage2) test_config4_dense_20k_*
"""
import types

import numpy as np
import pytest
import torch

from _parity import check_stage1_vs_fixture, oracle_trace, rel_err
from etch_amd.utils.weights import load_seeded

pytestmark = pytest.mark.gpu
ITEMS = ["confidence", "direction", "magnitude"]


def scan(seed, n, sigma=(0.14, 0.31, 0.085)):
    return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array(sigma)).astype(np.float32)


def make(tmp_path, seed=1):
    from etch_amd import constants as K
    from etch_amd.models.models_pointcloud import GT_network_equiv
    from etch_amd.utils.body_model import SyntheticSMPL
    args = types.SimpleNamespace(output_folder=str(tmp_path), EPN_input_radius=0.4, EPN_layer_num=2, device=torch.device("cuda"),
                                 markerset=K.default_markerset(), scale_magnitude=10, body_model=SyntheticSMPL(7))
    return args, load_seeded(GT_network_equiv(option=args), seed).cuda().eval()


def test_config0_bundled_scan_5k_vs_reference(tmp_path, golden):
    """The reference's bundled sample scan (datafolder/4D-DRESS/.../00122_Inner_Take2_00011.obj) centred and sampled like
    inference_demo.py:19-39 (seeded sampler), 5 000 points: real garment geometry, the real kernel-weight sparsity."""
    g, c = golden("scan_4ddress_5k.npz"), golden("constants.npz")
    args, model = make(tmp_path, int(g["seed"]))
    with torch.no_grad():
        res, sel = model(torch.from_numpy(g["points"]).cuda(), ITEMS, "standard_vector")
    assert sel.shape == (1, 5000, 3)
    print("config0 deviations:", check_stage1_vs_fixture(res, model.last_anc_w, g, c["anchors"]))


def test_config0_bundled_scan_full_pipeline_runs(tmp_path, golden):
    """configs[0] end to end on the same points: predict_smpl's call chain; the fit of THESE markers equals the oracle's."""
    from etch_amd.inference_demo import predict_smpl_batch
    from oracle import stage2 as S2
    g = golden("scan_4ddress_5k.npz")
    args, model = make(tmp_path, int(g["seed"]))
    meshes, markers, valid, info, aux = predict_smpl_batch(args, model, torch.from_numpy(g["points"]).cuda(), "neutral", return_trace=True)
    _check_fit_vs_oracle(args, markers.cpu(), valid.cpu(), aux, info)


def _check_fit_vs_oracle(args, markers, valid, aux, info, scans=None):
    """Stage 2 of the given scans against the oracle run on the SAME markers: per-iteration error trace, vertices, joints, V2V and
    every fitted parameter at 1e-4 (scans whose markers are NaN -- conf**20 underflow, as in the reference -- are skipped)."""
    from oracle import stage2 as S2
    scans = list(range(markers.shape[0])) if scans is None else list(scans)
    scans = [b for b in scans if bool(torch.isfinite(markers[b]).all())]
    assert scans, "every selected scan has NaN markers"
    mv = np.array(list(args.markerset.values()))
    trace = []
    ref = S2.fit_smpl(args.body_model, mv, markers[scans], valid[scans], trace=trace)
    rt = oracle_trace(trace)
    gt = aux["err_trace"].cpu().numpy()[scans]
    assert np.abs(gt - rt).max() / rt.max() < 1e-4
    verts = aux["verts"].cpu().numpy()[scans]
    assert np.abs(verts - ref["verts"].numpy()).max() < 1e-4
    assert np.abs(info[4][scans] - ref["joints"].numpy()).max() < 1e-4
    x = aux["x"].cpu().numpy()[scans]
    xr = torch.cat([ref["pose"], ref["betas"], ref["orient"], ref["transl"]], 1).numpy()
    dev = np.abs(x - xr)
    groups = {"body pose": slice(0, 63), "hands": slice(63, 69), "betas[:2]": slice(69, 71), "betas[2:]": slice(71, 79), "orient": slice(79, 82),
              "transl": slice(82, 85)}
    print("LM parameter deviation vs oracle:", {k: float(dev[:, s].max()) for k, s in groups.items()})
    assert dev.max() < 1e-4
    v2v = np.linalg.norm(verts - ref["verts"].numpy(), axis=-1).mean(1)
    assert v2v.max() < 1e-5                                                  # V2V (eval.py:235-237) of GPU vs oracle bodies: < 0.01 mm


def test_config1_2_batch32_5k_vs_reference_and_oracle(tmp_path, golden):
    """The metric's own workload: ONE batch of 32 x 5 000 synthetic scans (bench seeds 1000 + b).  Stage 1 of scans 0 and 31
    against the reference's Python (tests/golden/model_n5000.npz); stage 2 of the same scans against the oracle on the GPU's own
    markers.  Also checks that a scan's result does not depend on its batch neighbours (scan 0 alone == scan 0 in the batch)."""
    from etch_amd.inference_demo import predict_smpl_batch
    g, c = golden("model_n5000.npz"), golden("constants.npz")
    ids = g["scan_ids"].tolist()
    assert ids == [0, 31]
    args, model = make(tmp_path, int(g["seed"]))
    pts = np.stack([scan(1000 + b, 5000) for b in range(32)])
    assert np.array_equal(pts[ids], g["points"])
    dev = torch.from_numpy(pts).cuda()
    with torch.no_grad():
        res, _ = model(dev, ITEMS, "standard_vector")
        anc_w = model.last_anc_w.clone()
        print("config1 deviations (scans 0, 31 of 32):", check_stage1_vs_fixture(res, anc_w, g, c["anchors"], scans=ids))
        solo, _ = model(dev[:1].contiguous(), ITEMS, "standard_vector")
    for k in res:
        assert torch.equal(solo[k][0], res[k][0]), k
    meshes, markers, valid, info, aux = predict_smpl_batch(args, model, dev, "neutral", return_trace=True)
    assert len(meshes) == 32 and markers.shape == (32, 86, 3)
    _check_fit_vs_oracle(args, markers.cpu(), valid.cpu(), aux, info, scans=ids)


def test_config4_dense_20k_vs_reference(tmp_path, golden):
    """configs[4]'s geometry: one dense 20 000-point scan against the reference's Python (tests/golden/model_n20000.npz) -- the
    O(N^2) index kernels, the large-segment FPS variants and the spatial schedule beyond one LDS sort."""
    g, c = golden("model_n20000.npz"), golden("constants.npz")
    args, model = make(tmp_path, int(g["seed"]))
    with torch.no_grad():
        res, _ = model(torch.from_numpy(g["points"]).cuda(), ITEMS, "standard_vector")
    print("config4 geometry deviations:", check_stage1_vs_fixture(res, model.last_anc_w, g, c["anchors"]))


def test_padding_heavy_distribution_vs_reference_fp64(tmp_path, golden):
    """SURVEY 8d's second distribution (sigma = 0.20, 0.45, 0.12): a few points carry tokens ~12x the typical magnitude, their
    attention softmax saturates and amplifies fp32 rounding.  Evidence instead of argument: the fixture holds the REFERENCE Python
    run in fp64 on this input and, per point, how far the reference's OWN fp32 run lands from it (max 3.1e-4, 99.9 % < 2.4e-5).
    The GPU result must sit as close to the fp64 truth as the reference's fp32 run does (x2 for a different summation order)."""
    g = golden("padding_heavy_fp64.npz")
    args, model = make(tmp_path)
    with torch.no_grad():
        model(torch.from_numpy(g["points"]).cuda(), ["direction"], "standard_vector")
    dev = (model.last_anc_w.cpu().double().numpy() - g["anc_w_fp64"]).__abs__().max(-1) / float(g["scale"])
    ref = g["ref_fp32_dev"].astype(np.float64)
    q = lambda a, p: float(np.quantile(a.reshape(-1), p))
    print("padding-heavy anc_w deviation from the fp64 reference  gpu: max %.2e q99.9 %.2e median %.2e | reference fp32: max %.2e q99.9 %.2e median %.2e"
          % (dev.max(), q(dev, 0.999), q(dev, 0.5), ref.max(), q(ref, 0.999), q(ref, 0.5)))
    assert dev.max() <= 2.0 * ref.max()
    assert q(dev, 0.999) <= 2.0 * q(ref, 0.999) and q(dev, 0.5) <= 2.0 * q(ref, 0.5) + 1e-7
    assert (dev < 1e-4).mean() >= 0.999
