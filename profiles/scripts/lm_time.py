import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from test_gpu_stage2 import _problem
from etch_amd import ops
from etch_amd.models.fit_SMPL import _device_body
for model, B, it0, it1 in (("smpl", 32, 30, 50), ("smpl", 1, 30, 50), ("smplx", 8, 75, 125), ("smplx", 32, 75, 125)):
    bm, ms, mv, tgt, valid, _ = _problem(B, seed=3, model=model)
    db = _device_body(bm, mv, torch.device("cuda"))
    m, v = tgt.cuda(), valid.float().cuda()
    ph = torch.zeros(B, 8, dtype=torch.int64, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        x, x0, tr = ops.smpl_lm_fit(db.lm_consts, m, v, it0, 0.5, 0.01, it1, 0.2, 1e-3, True, phase_ticks=ph, nj=db.nj, nb=db.nb)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    trc = tr.cpu().numpy()
    frozen = (np.diff(trc, axis=1) == 0).sum(1)
    p = ph.cpu().numpy()[0].astype(float)
    if p[:5].sum() == 0: p[:] = np.nan
    print(f"{model} B={B} {it0}+{it1}: {dt*1e3:.2f} ms = {dt*1e6/(it0+it1+2):.1f} us/linearisation; frozen iterations per scan (mean) {frozen.mean():.1f}; "
          f"phase share kin {p[0]/p[:5].sum():.2f} markers+JtJ {p[1]/p[:5].sum():.2f} writeA {p[2]/p[:5].sum():.2f} chol {p[3]/p[:5].sum():.2f} backsub {p[4]/p[:5].sum():.2f}; of markers: A+A2+A3 {p[5]/p[1]:.2f} B {p[6]/p[1]:.2f} C {p[7]/p[1]:.2f} mfma {1-(p[5]+p[6]+p[7])/p[1]:.2f}; final err {trc[:, -1].mean():.3e}")
