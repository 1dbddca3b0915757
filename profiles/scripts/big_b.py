import sys, types
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench as BM
dev = torch.device("cuda")
args, model = BM.build(dev)
N = 5000
ref_pts = torch.from_numpy(np.stack([BM.synth_scan(i, N) for i in (0, 1)])).to(dev)
with torch.no_grad():
    ref, _ = model(ref_pts, ["confidence", "direction", "magnitude"])
ref = {k: v.clone() for k, v in ref.items()}
for B in (3, 100):
    pts = torch.from_numpy(np.stack([BM.synth_scan(i % 2 if i in (0, 1, B - 1) else 10 + i, N) for i in range(B)])).to(dev)
    with torch.no_grad():
        res, _ = model(pts, ["confidence", "direction", "magnitude"])
    torch.cuda.synchronize()
    ok = {k: (bool(torch.equal(res[k][0], ref[k][0])), bool(torch.equal(res[k][B - 1], ref[k][(B - 1) % 2])), bool(torch.isfinite(res[k].float()).all())) for k in ref}
    d = (res["confidences"][B - 1] - ref["confidences"][(B - 1) % 2]).abs()
    print(B, ok, "peak GiB %.1f" % (torch.cuda.max_memory_allocated() / 2**30), "confidence diff max %.3e at %d of %d, ref max %.3e, n differing %d" % (float(d.max()), int(d.argmax()), d.numel(), float(ref["confidences"].abs().max()), int((d > 0).sum())))
    del res, pts
