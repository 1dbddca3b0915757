"""Emit tests/golden/fit_conditioning.npz: a stage-2 CONDITIONING STRESS problem run by the oracle (oracle/stage2.py) in fp32 AND fp64.

SURVEY appendix C: at lambda = 1e-3 the LM normal matrix of a real SMPL fit has cond ~ 1e4 - 1e5 and fp32 rounding shows up in the weakly
observed DoFs (hands, feet, high betas).  The well-posed problems of the other stage-2 tests never get there (all 86 markers, 2 mm noise, gentle
blend shapes: their fp32 and fp64 oracles agree to 3e-6).  This one does:
  * blend shapes of realistic magnitude: shapedirs ~ 3e-2, posedirs ~ 1e-2 (SyntheticSMPL(7, shape_scale=0.03, pose_scale=0.01));
  * 15 - 25 valid markers per scan, NONE on hands / feet -- neither by marker name (…ANK, …HEE, …MT1/5, …TOE, …FIN, …THMB, …IWR, …OWR) nor by
    the marker vertex's dominant skinning joint (ankles 7/8, feet 10/11, wrists 20/21, hands 22/23): those joints are regularised by the
    damping alone;
  * 1 cm marker noise, the full 30 + 50 schedule.
Stored: the inputs, both oracle runs (parameters, error traces, a vertex subset, joints) and the condition number of the final damped normal
matrix.  The fp64 run is the rounding-free yardstick of the SAME algorithm; |fp32 run - fp64 run| per DoF group is the error any fp32
implementation is entitled to (the rule tests/_parity.py:check_stage1_vs_fixture applies to stage 1).  PARITY UNPINNED upstream as for every
LM / LBS fixture (no theseus / smplx / SMPL pickle in the reference tree).

    python -m oracle.gen_fit_conditioning_fixture
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from etch_amd import constants as K  # noqa: E402
from etch_amd.utils.body_model import SyntheticSMPL  # noqa: E402
from oracle import stage2 as S2  # noqa: E402

BODY = dict(seed=7, shape_scale=0.03, pose_scale=0.01)
EXTREMITY_JOINTS = (7, 8, 10, 11, 20, 21, 22, 23)
EXTREMITY_NAMES = ("ANK", "HEE", "MT1", "MT5", "TOE", "FIN", "THMB", "IWR", "OWR")


def problem(B=3, seed=21):
    bm = SyntheticSMPL(**BODY)
    ms = K.default_markerset()
    names = list(ms.keys())
    mv = np.array(list(ms.values()))
    dom = bm.lbs_weights[mv].argmax(1)
    allowed = np.array([not any(n[1:] == e or n.endswith(e) for e in EXTREMITY_NAMES) and dom[i] not in EXTREMITY_JOINTS for i, n in enumerate(names)])
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    tb = S2.TorchBody(bm)
    pose = torch.randn(B, 72, generator=g) * 0.25
    betas = torch.randn(B, 10, generator=g) * 1.0
    transl = torch.randn(B, 3, generator=g) * 0.05
    with torch.no_grad():
        vgt = S2.lbs(tb, betas, pose, transl)[0]
    tgt = vgt[:, mv] + torch.randn(B, 86, 3, generator=g) * 0.01
    valid = torch.zeros(B, 86, dtype=torch.bool)
    for b, k in enumerate((15, 20, 25)[:B]):
        valid[b, rng.choice(np.nonzero(allowed)[0], k, replace=False)] = True
    return bm, mv, tgt, valid, vgt, allowed


def final_condition(bm, mv, x, tgt, valid, damping=1e-3):
    tb = S2.TorchBody(bm, torch.float64)
    f = S2.residual_fn(tb, torch.as_tensor(mv).long(), 10)
    J = torch.func.vmap(torch.func.jacrev(f))(x.double(), tgt.double(), valid.double())
    A = J.transpose(1, 2) @ J + damping * torch.eye(J.shape[2], dtype=torch.float64)
    return torch.linalg.cond(A).numpy()


def main():
    torch.set_num_threads(8)
    bm, mv, tgt, valid, vgt, allowed = problem()
    runs = {}
    for tag, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
        trace = []
        fit = S2.fit_smpl(bm, mv, tgt, valid, trace=trace, dtype=dt)
        tr = np.concatenate([torch.stack(trace[0], 1).numpy(), torch.stack(trace[1], 1).numpy()], 1)
        x = torch.cat([fit["pose"], fit["betas"], fit["orient"], fit["transl"]], 1)
        runs[tag] = dict(x=x.numpy(), trace=tr, verts=fit["verts"][:, ::10].numpy(), joints=fit["joints"].numpy(), x_stage0=fit["x_stage0"].numpy())
    cond = final_condition(bm, mv, torch.from_numpy(runs["fp64"]["x"]), tgt, valid)
    d = np.abs(runs["fp32"]["x"].astype(np.float64) - runs["fp64"]["x"])
    groups = {"body pose": slice(0, 63), "hands": slice(63, 69), "betas[:2]": slice(69, 71), "betas[2:]": slice(71, 79), "orient": slice(79, 82),
              "transl": slice(82, 85)}
    print("cond(J^T J + 1e-3 I) at the fp64 solution:", cond)
    print("oracle fp32 vs fp64 parameter deviation:", {k: float(d[:, s].max()) for k, s in groups.items()})
    print("trace lengths:", {t: r["trace"].shape for t, r in runs.items()}, "final err fp64", runs["fp64"]["trace"][:, -1])
    out = os.path.join(ROOT, "tests", "golden", "fit_conditioning.npz")
    np.savez_compressed(out, body=json.dumps(BODY), markers=tgt.numpy(), valid=valid.numpy(), allowed=allowed, cond=cond,
                        **{f"{k}_{t}": v for t, r in runs.items() for k, v in r.items()})
    print("wrote", out, os.path.getsize(out) / 1e3, "kB")


if __name__ == "__main__":
    main()
