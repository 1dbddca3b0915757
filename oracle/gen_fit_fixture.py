"""Emit tests/golden/fit_oracle.npz: a pinned run of the oracle's stage-2 restatement (oracle/stage2.py) on the seeded
SMPL-shaped body model -- inputs (noisy marker targets, valid mask), the per-iteration LM error trace of both stages,
the final parameters, joints and a vertex subset.  PARITY UNPINNED upstream (no theseus / smplx / SMPL pickle in the
reference tree or the image): this fixture pins the ORACLE against regressions and gives the GPU tests a fast target.

    python -m oracle.gen_fit_fixture
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from etch_amd import constants as K  # noqa: E402
from etch_amd.utils.body_model import SyntheticSMPL  # noqa: E402
from oracle import stage2 as S2  # noqa: E402


def problem(B, seed=0):
    bm = SyntheticSMPL(7)
    mv = np.array(list(K.default_markerset().values()))
    tb = S2.TorchBody(bm)
    g = torch.Generator().manual_seed(seed)
    gt_pose = torch.randn(B, 72, generator=g) * 0.2
    gt_b = torch.randn(B, 10, generator=g) * 0.8
    gt_t = torch.randn(B, 3, generator=g) * 0.05
    with torch.no_grad():
        vgt = S2.lbs(tb, gt_b, gt_pose, gt_t)[0]
    tgt = vgt[:, mv] + torch.randn(B, 86, 3, generator=g) * 0.002
    valid = torch.ones(B, 86, dtype=torch.bool)
    valid[0, 5] = False
    valid[B - 1, 40:44] = False
    return bm, mv, tgt, valid, vgt


def main():
    torch.set_num_threads(8)
    bm, mv, tgt, valid, vgt = problem(2, seed=11)
    trace = []
    fit = S2.fit_smpl(bm, mv, tgt, valid, trace=trace)
    tr = torch.cat([torch.stack(trace[0], 1), torch.stack(trace[1], 1)], 1)
    x = torch.cat([fit["pose"], fit["betas"], fit["orient"], fit["transl"]], 1)
    out = os.path.join(ROOT, "tests", "golden", "fit_oracle.npz")
    np.savez_compressed(out, body_seed=7, problem_seed=11, markers=tgt.numpy(), valid=valid.numpy(), err_trace=tr.numpy(), x=x.numpy(),
                        x_stage0=fit["x_stage0"].numpy(), joints=fit["joints"].numpy(), verts_sub=fit["verts"][:, ::10].numpy(),
                        v2v_vs_generating=(fit["verts"] - vgt).norm(dim=-1).mean(1).numpy())
    print("wrote", out, os.path.getsize(out) / 1e3, "kB; final err", tr[:, -1].tolist())


if __name__ == "__main__":
    main()
