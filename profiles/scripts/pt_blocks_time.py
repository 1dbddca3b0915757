"""A/B timing of the Point-Transformer block runs of the bench workload (B = 32 x 5 000 points): fused K1/K2 kernels vs the four-kernel
blocks, per level.  python profiles/scripts/pt_blocks_time.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from etch_amd.models import pointops  # noqa: E402
from etch_amd.models import pointtransformer_seg as P  # noqa: E402
from etch_amd.utils.weights import load_seeded  # noqa: E402

B = 32
LEVELS = [(5000, 64, 8, 1), (5000, 128, 8, 1), (1250, 128, 16, 2), (312, 256, 16, 3), (78, 256, 16, 5), (19, 512, 16, 2), (312, 256, 16, 1), (78, 256, 16, 1)]


def timeit(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for per, c, ns, nb in LEVELS:
    n = per * B
    rng = np.random.default_rng(per + c)
    p = torch.from_numpy((rng.standard_normal((n, 3)) * np.array([0.14, 0.31, 0.085])).astype(np.float32)).cuda()
    x = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).cuda()
    o = pointops.offsets_tensor([per * (i + 1) for i in range(B)], "cuda")
    blocks = [load_seeded(P.PointTransformerBlock(c, c, 8, ns), 3 + k).cuda().eval() for k in range(nb)]
    with torch.no_grad(), pointops.knn_scope():
        P.PointTransformerBlock.fused = True
        tf = timeit(lambda: P.run_blocks(blocks, [p, x, o]))
        P.PointTransformerBlock.fused = False
        tu = timeit(lambda: P.run_blocks(blocks, [p, x, o]))
    print(f"{per:5d} pts/scan  c={c:3d} ns={ns:2d} blocks={nb}:  fused {tf:8.1f} us   4-kernel {tu:8.1f} us   ({tf / tu:.2f}x)", flush=True)
