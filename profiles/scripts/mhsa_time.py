"""The two attention layers of the direction head at 32 x 5000 points: ms per launch (HIP events)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from etch_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
T = 160000
x = torch.randn(T * 60, 64, generator=g).to(dev)
W = [(torch.randn(64, 64, generator=g) * 0.125).to(dev) for _ in range(4)]
bc = (torch.randn(64, generator=g) * 0.1).to(dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
a = ops.mhsa_layer(x, W[0], W[1], W[2], W[3], bc, mode=0)
b = ops.mhsa_layer(x, W[0], W[1], W[2], mode=2)
print("mode 0 %.3f ms   mode 2 %.3f ms   checksums %.6e %.6e" % (t(lambda: ops.mhsa_layer(x, W[0], W[1], W[2], W[3], bc, mode=0)), t(lambda: ops.mhsa_layer(x, W[0], W[1], W[2], mode=2)), float(a.double().sum()), float(b.double().sum())))
