"""Times every etch_inter_so3conv / etch_intra_so3conv call of one encoder pass at B=32 x 5000 (HIP events, one stream)."""
import collections, os, sys, types
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
        model.encoder(pts)
for _ in range(2): run()
torch.cuda.synchronize()
tot = collections.OrderedDict()
for rep in range(3):
    agg = BM.profile_pass(run)
    for k, v in agg.items():
        if "so3conv" in k:
            tot.setdefault(k, []).append(v["ms"])
print(os.environ.get("ETCH_HIP_LIB", "default").split("libetch_")[-1], "mfma32=" + os.environ.get("ETCH_INTER_MFMA32", "1"), " ".join(f"{k.replace('_so3conv_kernel','')}={min(v):.3f}" for k, v in tot.items()),
      "inter_sum=%.3f" % sum(min(v) for k, v in tot.items() if k.startswith("inter_so3conv")))
