"""One encoder pass at B=32 x 5000 (for PMC collection)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench as BM
dev = torch.device("cuda")
args, model = BM.build(dev)
model.concurrent_heads, model.overlap_index_ops = False, False
pts = torch.from_numpy(np.stack([BM.synth_scan(i, 5000) for i in range(32)])).to(dev)
with torch.no_grad():
    model.encoder(pts)
torch.cuda.synchronize()
