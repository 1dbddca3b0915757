#!/usr/bin/env python3
"""CPU baseline beside the GPU number: the oracle (torch-CPU restatement of both stages) on the box's host cores at
1 / 32 / all torch threads, one scan and a batch of four.  Test infrastructure timed as a baseline, never the product.

    python profiles/cpu_baseline_sweep.py > profiles/r02_cpu_baseline_threads.txt
"""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BM  # noqa: E402
from etch_amd import constants as K  # noqa: E402
from etch_amd.models.models_pointcloud import GT_network_equiv  # noqa: E402
from etch_amd.utils.body_model import SyntheticSMPL  # noqa: E402
from etch_amd.utils.weights import seeded_state_dict  # noqa: E402
from oracle import stage1 as S1  # noqa: E402
from oracle import stage2 as S2  # noqa: E402

ms = K.default_markerset()
args = types.SimpleNamespace(output_folder="/tmp/etch_cpu_sweep", EPN_input_radius=0.4, EPN_layer_num=2, device=torch.device("cpu"), markerset=ms)
sd = seeded_state_dict(GT_network_equiv(option=args), 1)
table = S1.build_layer_table()
bm = SyntheticSMPL(7)
mv = np.array(list(ms.values()))
print(f"# host: {os.cpu_count()} hardware threads; oracle stage 1 + get_markers + 30+50-iteration autograd LM, 5 000-point synthetic scans (bench seeds)")
print(f"{'threads':>8s} {'batch':>6s} {'stage1_s':>10s} {'stage2_s':>10s} {'scans_per_s':>12s}")
for threads, batch in ((1, 1), (8, 1), (32, 1), (32, 4), (os.cpu_count() // 2, 1), (os.cpu_count() // 2, 4)):
    torch.set_num_threads(threads)
    x = torch.from_numpy(np.stack([BM.synth_scan(i, 5000) for i in range(batch)]))
    t0 = time.time()
    out = S1.forward(sd, x, table, num_markers=len(ms))
    t1 = time.time() - t0
    labels = out["part_labels"].argmax(-1)
    inner = x - out["direction"] * out["magnitude"] / 10
    t0 = time.time()
    mk, valid = S2.get_markers(len(ms), inner, labels, out["confidences"])
    S2.fit_smpl(bm, mv, mk, valid)
    t2 = time.time() - t0
    print(f"{threads:8d} {batch:6d} {t1:10.1f} {t2:10.1f} {batch / (t1 + t2):12.4f}", flush=True)
