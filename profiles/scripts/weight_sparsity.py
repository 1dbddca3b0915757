"""How many (point, anchor, neighbour) triples of the inter convs have any non-zero kernel weight (CPU, oracle functions only)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import stage1 as S
from etch_amd import constants as C
anchors = torch.from_numpy(C._c()['anchors']).float()
fx = np.load('tests/golden/scan_4ddress_5k.npz')
def run(pts, name):
    xyz = torch.from_numpy(pts).float().t()[None].contiguous()
    table = S.build_layer_table()
    for bi, blk in enumerate(table):
        for ci, cfg in enumerate(blk):
            g, ball, sidx, new_xyz = S.inter_grouping(xyz, cfg['stride'], cfg['radius'], cfg['n_neighbor'], cfg['lazy_sample'])
            kern = torch.from_numpy(C.get_kernel_points(cfg['radius']))
            p2 = g.shape[2]; nn = g.shape[3]
            sel = slice(0, min(p2, 400))
            w = S.inter_weights(g[:, :, sel], anchors, kern, cfg['sigma'])[0]      # p, na, ks, nn
            nz = (w > 0)
            anyk = nz.any(2)                      # p, na, nn
            # distinct neighbours: the ball query pads with the first index; count padded slots separately
            b0 = ball[0, sel].numpy()
            valid_cnt = np.array([len(np.unique(r)) for r in b0])
            per = anyk.sum(2).float()             # useful neighbours per (p, a)
            per4 = (torch.ceil(per / 4) * 4)
            print(f"{name} b{bi}c{ci} cin={cfg['dim_in']} nn={nn} p2={p2}: weights nonzero {nz.float().mean():.3f}; neighbours with any nonzero k {anyk.float().mean():.3f}; "
                  f"mean useful per (p,a) {per.mean():.1f} -> padded to 4: {per4.mean():.1f} of {nn} ({per4.mean()/nn:.3f}); distinct neighbours {valid_cnt.mean():.1f}")
            xyz = new_xyz
run(fx['points'].reshape(-1,3)[:5000], '4ddress')
import bench
run(bench.synth_scan(0, 5000), 'bench')
