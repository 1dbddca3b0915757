"""Times the three inter convs of the released encoder depth at the bench batch (32 x 5 000 points): round-3 kernel (step 1 on the fp32 MFMA),
round-4 planes kernels, round-5 kq kernel (weights' pre-activation on the matrix cores).  python profiles/scripts/time_inter_kq.py [reps] [only_kq]"""
import os
import sys

import torch

sys.path.insert(0, ".")
from etch_amd import ops  # noqa: E402
from etch_amd import vgtk_so3conv as V  # noqa: E402
from etch_amd.utils.weights import load_seeded  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only_kq = len(sys.argv) > 2
B = 32
g = torch.Generator().manual_seed(0)
pts = (torch.randn(B, 5000, 3, generator=g) * torch.tensor([0.14, 0.31, 0.085])).cuda()
xyz0 = pts.permute(0, 2, 1).contiguous()
fps = ops.furthest_point_sampling(xyz0, 2500)
xyz1 = ops.gather_points_forward(xyz0, fps)
shapes = [("b0c1", 32, 32, 32, 0.113137, 0.0064, xyz1, 2500), ("b1c0", 32, 64, 64, 0.16, 0.0128, xyz1, 1250), ("b1c1", 64, 64, 32, 0.16, 0.0128, xyz1[:, :, :1250].contiguous(), 1250)]
line = [os.environ.get("ETCH_HIP_LIB", "default").split("libetch_")[-1]]
tot = 0.0
for name, cin, cout, nn, radius, sigma, xyz, p2 in shapes:
    new_xyz = xyz[:, :, :p2].contiguous()
    ball = ops.ball_query(new_xyz, xyz, radius, nn)
    conv = load_seeded(V.InterSO3Conv(cin, cout, 1, 1, radius, sigma, nn), 3).cuda()
    rk, W, Wp, bias = conv._derived()
    feats = torch.randn(B, xyz.shape[2], 60, cin, generator=g).cuda()
    planes = ops.split3_planes(feats)
    order = ops.spatial_order(new_xyz)
    wq32 = ops.inter_weight_split32(W, cin)
    forms = [("kq", dict(Wqh=ops.inter_weight_split32_f16(W, cin), kq=ops.inter_kpoint_operand(rk, sigma), feats_planes=ops.split2_planes_f16(feats)))]
    if not only_kq:
        forms = [("r03", dict(Wq=conv._wq())), ("r04", dict(Wq32=wq32, feats_planes=planes) if cin == 64 else dict(Wqn=ops.inter_weight_split(W, cin, natural=True), feats_planes=planes))] + forms
    res = {}
    for label, kw in forms:
        f = lambda: ops.inter_so3conv(xyz, new_xyz, ball, feats, rk, W, Wp, bias, sigma, order=order, want_stats=True, **kw)
        y = f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[label] = (e0.elapsed_time(e1) / reps, y[0])
    tot += res["kq"][0]
    if only_kq:
        # checksum of the output bits and of the statistics (A/B of two builds of the kernel: equal checksums = bitwise equal results)
        cs = int(res["kq"][1].view(torch.int32).to(torch.int64).sum()) & 0xffffffffffff
        line.append(f"{name} {res['kq'][0]:.3f} [{cs:012x}]")
    else:
        d = float((res["kq"][1] - res["r03"][1]).abs().max()) / float(res["r03"][1].abs().max())
        print(f"{name} {cin}->{cout} nn={nn} p2={p2}: r03 {res['r03'][0]:.3f} ms, r04 {res['r04'][0]:.3f} ms, kq {res['kq'][0]:.3f} ms, kq vs r03 max diff / scale {d:.2e}", flush=True)
if only_kq:
    print(" ".join(line), f"sum {tot:.3f}", flush=True)
