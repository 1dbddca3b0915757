import sys, types
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench as BM
from etch_amd.inference_demo import predict_smpl_batch
dev = torch.device("cuda")
args, model = BM.build(dev)
N = 1024
rng = np.random.default_rng(0)
base = (rng.standard_normal((N, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)
cases = {
    "identical points": np.tile(base[:1], (N, 1)),
    "two clusters of duplicates": np.concatenate([np.tile(base[:1], (N // 2, 1)), np.tile(base[1:2], (N // 2, 1))]),
    "collinear": np.stack([np.linspace(-0.3, 0.3, N), np.zeros(N), np.zeros(N)], 1).astype(np.float32),
    "planar lattice": np.stack(np.meshgrid(np.linspace(-0.2, 0.2, 32), np.linspace(-0.4, 0.4, 32)), -1).reshape(-1, 2).astype(np.float32) @ np.array([[1, 0, 0], [0, 1, 0]], np.float32),
    "one NaN point": np.concatenate([base[:-1], np.full((1, 3), np.nan, np.float32)]),
    "huge coordinates": base * 1e6,
    "tiny extent": base * 1e-6,
}
for name, pts in cases.items():
    x = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)[None]).to(dev)
    try:
        with torch.no_grad():
            res, _ = model(x, ["confidence", "direction", "magnitude"])
        torch.cuda.synchronize()
        fin = {k: float(torch.isfinite(v.float()).float().mean()) for k, v in res.items()}
        meshes, markers, valid, info = predict_smpl_batch(args, model, x)
        print(f"{name:28s} finite share {fin}  valid markers {int(valid.sum())}  verts finite {bool(np.isfinite(meshes[0].vertices).all())}")
    except Exception as e:
        print(f"{name:28s} raised {type(e).__name__}: {str(e)[:100]}")
