"""Direction head: fused interpolation (mhsa_project + mhsa_interp_layer) vs prop_interp + mhsa_layer; correctness + time."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from etch_amd import ops
dev = torch.device('cuda:0')
B, N, S = 32, 5000, 1250
g = torch.Generator().manual_seed(0)
import bench
pts = torch.stack([torch.from_numpy(bench.synth_scan(i, N)) for i in range(B)]).to(dev)
# coarse points = a subset of the scan
sel = torch.stack([torch.randperm(N, generator=g)[:S] for _ in range(B)]).to(dev)
xyz2 = torch.gather(pts, 1, sel[..., None].expand(-1, -1, 3)).permute(0, 2, 1).contiguous()
F = torch.randn(B, S, 60, 64, generator=g).to(dev)
W = [(torch.randn(64, 64, generator=g) * 0.125).to(dev) for _ in range(4)]
bc = (torch.randn(64, generator=g) * 0.1).to(dev)
idx, w = ops.prop3nn(pts, xyz2)
order = ops.spatial_order(pts.permute(0, 2, 1).contiguous())
def unfused():
    x, inv = ops.prop_interp(F, idx, w, order=order)
    return ops.mhsa_layer(x.view(-1, 64), W[0], W[1], W[2], W[3], bc, mode=0), inv
def fused(od=order):
    cm = ops.token_mean(F.view(B * S, 60, 64))
    _, inv = ops.prop_interp(cm.view(B, S, 1, 64), idx, w, order=od)
    return ops.mhsa_interp_layer(F, idx, w, W[0], W[1], W[2], W[3], bc, order=od), inv
a, ia = unfused(); b, ib = fused()
if len(sys.argv) > 1 and sys.argv[1] == "once":
    torch.cuda.synchronize(); sys.exit(0)
torch.cuda.synchronize()
print("max rel err layer", float((a - b).abs().max() / a.abs().max()), "inv", float((ia - ib).abs().max() / ia.abs().max()))
b2, _ = fused(None)
print("order-independent bitwise", bool(torch.equal(b, b2)))
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print("unfused %.3f ms  fused %.3f ms  fused without order %.3f ms" % (t(unfused), t(fused), t(lambda: fused(None))))
print("interp layer %.3f  layer0 alone %.3f  prop_interp %.3f" % (t(lambda: ops.mhsa_interp_layer(F, idx, w, W[0], W[1], W[2], W[3], bc, order=order)), t(lambda: ops.mhsa_layer(a, W[0], W[1], W[2], W[3], bc, mode=0)),
      t(lambda: ops.prop_interp(F, idx, w, order=order))))
