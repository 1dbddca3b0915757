"""Tile-level sparsity of the kernel weights: with the 24 kernel points split 16 + 8 (second MFMA column tile = a compact cap of 8),
how many neighbours of a (point, anchor) have all-zero weights on the cap?"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import stage1 as S
from etch_amd import constants as C
import bench
anchors = torch.from_numpy(C._c()['anchors']).float()
kp = C._c()['kp24_raw'].astype(np.float64)
# cap: the 8 kernel points nearest to a direction; try all 23 shell points as the pole and keep the most compact
best = None
for pole in range(1, 24):
    d = np.linalg.norm(kp - kp[pole], axis=1); d[0] = 9
    cap = np.argsort(d)[:8]
    spread = d[cap].max()
    if best is None or spread < best[0]: best = (spread, cap)
cap = np.sort(best[1]); rest = np.array([k for k in range(24) if k not in cap])
print('cap', cap, 'spread', best[0])
def run(pts, name):
    xyz = torch.from_numpy(pts).float().t()[None].contiguous()
    for bi, blk in enumerate(S.build_layer_table()):
        for ci, cfg in enumerate(blk):
            g, ball, sidx, new_xyz = S.inter_grouping(xyz, cfg['stride'], cfg['radius'], cfg['n_neighbor'], cfg['lazy_sample'])
            kern = torch.from_numpy(C.get_kernel_points(cfg['radius']))
            nn = g.shape[3]
            w = S.inter_weights(g[:, :, :300], anchors, kern, cfg['sigma'])[0]      # p, na, ks, nn
            nz = w > 0
            capnz = nz[:, :, cap].any(2)          # p, na, nn
            restnz = nz[:, :, rest].any(2)
            ncap = capnz.sum(2).float(); nrest = restnz.sum(2).float()
            m_now = 2 * nn / 4
            m_new = torch.ceil(ncap / 4) + torch.ceil(nrest / 4)
            # 3-way: 8+8+8 tiles?  each tile a compact group
            print(f"{name} b{bi}c{ci} nn={nn}: cap non-zero for {capnz.float().mean():.3f} of neighbours, rest non-zero {restnz.float().mean():.3f}; "
                  f"MFMAs per (p,a): {m_now:.0f} -> {m_new.mean():.1f} ({m_new.mean()/m_now:.3f})")
            xyz = new_xyz
fx = np.load('tests/golden/scan_4ddress_5k.npz')
run(fx['points'].reshape(-1, 3)[:5000], '4ddress')
run(bench.synth_scan(0, 5000), 'bench')
