import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from etch_amd import ops, _lib
import ctypes
dev = torch.device("cuda")
def scan(seed, n): return (np.random.default_rng(seed).standard_normal((n, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)
B = 32
for ns, ms, k in ((5000, 5000, 8), (5000, 5000, 16), (5000, 1250, 16), (1250, 5000, 3), (1250, 1250, 16), (313, 313, 16), (79, 79, 16)):
    xyz = torch.from_numpy(np.concatenate([scan(i, ns) for i in range(B)])).to(dev)
    q = torch.from_numpy(np.concatenate([scan(i, ns)[:ms] if ms <= ns else scan(100 + i, ms) for i in range(B)])).to(dev)
    off = torch.tensor([ns * (i + 1) for i in range(B)], dtype=torch.int32, device=dev)
    noff = torch.tensor([ms * (i + 1) for i in range(B)], dtype=torch.int32, device=dev)
    idx = torch.empty((B * ms, k), dtype=torch.int32, device=dev); d = torch.empty((B * ms, k), dtype=torch.float32, device=dev)
    def run():
        _lib.check(_lib.lib().etch_knnquery(B, ms, B * ms, k, ctypes.c_void_p(xyz.data_ptr()), ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(off.data_ptr()),
                                            ctypes.c_void_p(noff.data_ptr()), ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(d.data_ptr()), 0,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "knn")
    run(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run()
    e.record(); torch.cuda.synchronize()
    print(f"support {ns} queries {ms} k {k}: {s.elapsed_time(e) / 5 * 1e3:8.1f} us   checksum {int(idx.long().sum())}")
