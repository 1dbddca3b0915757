"""Emit tests/golden/fit_smplx_handover.npz: the 188-DoF (SMPL-X-sized, 55 joints, 20 coefficients) LM fit of the oracle (oracle/stage2.py)
on well-posed markers (all 86 but a few masked, 2 mm noise) over 75 + 10 iterations -- every stage-0 iteration of BASELINE configs[4]'s
schedule and the hand-over into the 20-coefficient stage -- in fp32 AND fp64.  30 finger joints are observed by few markers, so raw
parameters carry fp32 noise of ~1e-4 at lambda = 1e-3; the fp64 run is the yardstick (see gen_fit_conditioning_fixture.py).  The oracle
differentiates the full 10 475-vertex mesh (~0.4 s per iteration and scan), which is why this is a committed fixture and not run inside
the GPU test.  PARITY UNPINNED upstream (no theseus / smplx / SMPL-X model in the reference tree).

    python -m oracle.gen_fit_smplx_fixture
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from etch_amd import constants as K  # noqa: E402
from etch_amd.utils.body_model import SyntheticSMPLX  # noqa: E402
from oracle import stage2 as S2  # noqa: E402

IT = (75, 10)


def problem(B=2, seed=6):
    bm = SyntheticSMPLX(7)
    mv = np.array(list(K.default_markerset().values()))
    tb = S2.TorchBody(bm)
    g = torch.Generator().manual_seed(seed)
    gt_pose = torch.randn(B, 3 * bm.num_joints, generator=g) * 0.2
    gt_b = torch.randn(B, bm.num_betas, generator=g) * 0.8
    gt_t = torch.randn(B, 3, generator=g) * 0.05
    with torch.no_grad():
        vgt = S2.lbs(tb, gt_b, gt_pose, gt_t)[0]
    tgt = vgt[:, mv] + torch.randn(B, 86, 3, generator=g) * 0.002
    valid = torch.ones(B, 86, dtype=torch.bool)
    valid[0, 5] = False
    valid[B - 1, 40:44] = False
    return bm, mv, tgt, valid


def main():
    torch.set_num_threads(8)
    bm, mv, tgt, valid = problem()
    out = dict(markers=tgt.numpy(), valid=valid.numpy(), iters=np.array(IT))
    for tag, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
        trace = []
        fit = S2.fit_smpl(bm, mv, tgt, valid, steps_stage0=IT[0], steps_stage1=IT[1], trace=trace, dtype=dt)
        tr = []
        for t, n in zip(trace, (IT[0] + 1, IT[1] + 1)):
            t = torch.stack(t, 1)
            if t.shape[1] < n:
                t = torch.cat([t, t[:, -1:].expand(-1, n - t.shape[1])], 1)
            tr.append(t.numpy())
        x = torch.cat([fit["pose"], fit["betas"], fit["orient"], fit["transl"]], 1)
        out.update({f"x_{tag}": x.numpy(), f"trace_{tag}": np.concatenate(tr, 1), f"verts_{tag}": fit["verts"][:, ::10].numpy(),
                    f"joints_{tag}": fit["joints"].numpy(), f"x_stage0_{tag}": fit["x_stage0"].numpy()})
    d = np.abs(out["x_fp32"].astype(np.float64) - out["x_fp64"])
    print("oracle fp32 vs fp64: pose %.2e betas %.2e orient %.2e transl %.2e; verts %.2e" %
          (d[:, :162].max(), d[:, 162:182].max(), d[:, 182:185].max(), d[:, 185:].max(), np.abs(out["verts_fp32"] - out["verts_fp64"]).max()))
    path = os.path.join(ROOT, "tests", "golden", "fit_smplx_handover.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e3, "kB")


if __name__ == "__main__":
    main()
