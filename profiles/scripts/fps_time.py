import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from etch_amd import ops
def scan(seed, n): return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)
for b, n, m in ((32, 5000, 2500), (8, 20000, 10000), (1, 5000, 2500), (32, 1250, 312)):
    x = torch.from_numpy(np.ascontiguousarray(np.stack([scan(1000 + i, n).T for i in range(b)]))).cuda()
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        idx = ops.furthest_point_sampling(x, m)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"vgtk FPS b={b} {n}->{m}: {dt*1e3:.2f} ms = {dt*1e6/(m-1):.3f} us/round")
