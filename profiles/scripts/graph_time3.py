import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench as BM
from etch_amd.graph import GraphedHotPath
from etch_amd.inference_demo import predict_smpl_batch
from etch_amd.pipeline import HotPathPipeline
dev = torch.device("cuda")
args, model = BM.build(dev)
N = 5000
def t(fn, n=15):
    fn(); torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / n * 1e3
pts = torch.from_numpy(BM.synth_scan(777, N)[None]).to(dev)
print("fresh: eager %.2f" % t(lambda: predict_smpl_batch(args, model, pts)))
big = [torch.from_numpy(np.stack([BM.synth_scan(100 * k + i, N) for i in range(32)])).to(dev) for k in range(4)]
which = sys.argv[1] if len(sys.argv) > 1 else "pipe"
if which == "pipe":
    pipe = HotPathPipeline(args, model, "neutral", max_in_flight=2)
    for _ in pipe.run(iter(big)): pass
elif which == "eager32":
    for b in big: predict_smpl_batch(args, model, b)
torch.cuda.synchronize()
print("after %s: eager %.2f" % (which, t(lambda: predict_smpl_batch(args, model, pts))))
g = GraphedHotPath(args, model, 1, N)
print("graph %.2f  device-only %.2f  eager %.2f" % (t(lambda: g(pts)), t(lambda: g.replay(pts)), t(lambda: predict_smpl_batch(args, model, pts))))
