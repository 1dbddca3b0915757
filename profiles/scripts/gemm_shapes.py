"""Per-shape time of every etch_linear / pt attention call of one serial forward at B=32 x 5000."""
import collections, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench as BM
from etch_amd import _lib
dev = torch.device("cuda")
args, model = BM.build(dev)
model.concurrent_heads, model.overlap_index_ops = False, False
pts = torch.from_numpy(np.stack([BM.synth_scan(i, 5000) for i in range(32)])).to(dev)
def run():
    with torch.no_grad():
        model(pts, ["confidence", "direction", "magnitude"])
for _ in range(2): run()
torch.cuda.synchronize()
rec = []
def prof(name, a, fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); st = fn(*a); e.record(); rec.append((name, a, s, e)); return st
lib = _lib.lib(); lib.profiler = prof
run(); torch.cuda.synchronize(); lib.profiler = None
agg = collections.OrderedDict(); tot = collections.Counter()
for name, a, s, e in rec:
    v = [x.value if hasattr(x, "value") else x for x in a]
    ms = s.elapsed_time(e)
    tot[name] += ms
    if name == "etch_linear": key = (name, v[0], v[1], v[2])
    elif name.startswith("etch_pt_attention"): key = (name, v[0], v[1], v[2])
    else: continue
    d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += ms
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    fl = 2.0 * k[1] * k[2] * k[3] if k[0] == "etch_linear" else 0
    print(f"{k[0]:24s} R={k[1]:7d} K/c={k[2]:4d} O/ns={k[3]:4d} calls {n:3d} total {ms:7.3f} ms  each {ms/n*1e3:7.1f} us" + (f"  {fl/ (ms/n*1e-3)/1e12:5.1f} TF/s" if fl else ""))
print({k: round(v, 2) for k, v in tot.most_common(14)})
