"""Round time of the FPS kernels, one workgroup per scan against the split over G workgroups (ops.furthest_point_sampling(split=G)).
   python profiles/scripts/fps_time.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from etch_amd import ops
def scan(seed, n): return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)
cus = torch.cuda.get_device_properties(0).multi_processor_count
for b, n, m in ((32, 5000, 2500), (1, 5000, 2500), (64, 20000, 10000), (8, 20000, 10000), (1, 20000, 10000), (64, 20000, 5000), (32, 10000, 5000)):
    x = torch.from_numpy(np.ascontiguousarray(np.stack([scan(1000 + i, n).T for i in range(b)]))).cuda()
    ref = None
    import ctypes
    from etch_amd import _lib
    _lib.lib().etch_fps_fast(0)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        idx0 = ops.furthest_point_sampling(x, m, split=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    _lib.lib().etch_fps_fast(1)
    print(f"vgtk FPS b={b:3d} {n}->{m}: G=1, 64-bit-key slot loop  {dt*1e3:7.2f} ms = {dt*1e6/(m-1):.3f} us/round", flush=True)
    for G in (1, 2, 4, 8):
        if G > 1 and (b * G > cus or n > 8192 * G):
            continue
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            idx = ops.furthest_point_sampling(x, m, split=G)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        if ref is None:
            ref = idx0
        same = bool(torch.equal(ref, idx))
        print(f"vgtk FPS b={b:3d} {n}->{m}: G={G}  {dt*1e3:7.2f} ms = {dt*1e6/(m-1):.3f} us/round  identical={same}  auto G={ops.fps_split_default(b, n)}", flush=True)
