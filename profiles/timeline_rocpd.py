#!/usr/bin/env python3
"""Concurrency summary of a rocprofv3 kernel trace (rocpd sqlite): for the window spanned by the timed pipelined steps
(between the first and last smpl_lm_fit_kernel on the dedicated stage-2 stream) print, per HIP stream / HSA queue,
the kernel-busy time, the union busy time of the device, and the largest idle gaps of the busiest queue.

    python profiles/timeline_rocpd.py gpurun_out/prof/NAME_results.db
"""
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    lm = [r for r in rows if "smpl_lm_fit_kernel" in r[0]]
    # pipelined steps: the fits that ran on the dedicated stage-2 stream (the stream that carries nothing but the fit, the final
    # LBS and copies); fits of synchronous steps run on the stream that also carries the network
    names = {}
    for r in rows:
        names.setdefault(r[4], set()).add(r[0].split("(")[0])
    s2 = [sid for sid in {r[4] for r in lm} if len(names[sid]) <= 8]
    if s2:
        lm = [r for r in lm if r[4] == s2[0]]
    t0, t1 = lm[0][1], lm[-1][1]
    nsteps = len(lm) - 1
    win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    wall = t1 - t0
    print(f"# window: {nsteps} pipelined steps, {wall / 1e6:.2f} ms  ({wall / nsteps / 1e6:.2f} ms/step), {len(win)} kernels")
    print(f"device busy (union of all kernels): {union([(r[1], r[2]) for r in win]) / wall * 100:.1f} % of the window")
    by = {}
    for r in win:
        by.setdefault((r[3], r[4]), []).append(r)
    print(f"{'queue':>6s} {'stream':>7s} {'kernels':>8s} {'busy ms/step':>13s} {'busy %':>7s}  top kernels")
    for key, rs in sorted(by.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        busy = sum(r[2] - r[1] for r in rs)
        agg = {}
        for r in rs:
            agg[r[0].split("(")[0][:40]] = agg.get(r[0].split("(")[0][:40], 0) + r[2] - r[1]
        top = ", ".join(f"{k} {v / nsteps / 1e6:.1f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:3])
        print(f"{key[0]:6d} {key[1]:7d} {len(rs):8d} {busy / nsteps / 1e6:13.2f} {busy / wall * 100:7.1f}  {top}")
    key, rs = max(by.items(), key=lambda kv: sum(r[2] - r[1] for r in kv[1]))
    rs = sorted(rs, key=lambda r: r[1])
    gaps = sorted(((rs[i + 1][1] - rs[i][2], rs[i][0].split("(")[0][:36], rs[i + 1][0].split("(")[0][:36]) for i in range(len(rs) - 1)), reverse=True)
    tot_gap = sum(g[0] for g in gaps if g[0] > 0)
    print(f"busiest stream: idle between its kernels {tot_gap / nsteps / 1e6:.2f} ms/step; largest gaps (us, after -> before):")
    for g in gaps[:8]:
        print(f"  {g[0] / 1e3:9.1f}  {g[1]} -> {g[2]}")
    # the step boundary of the stage-1 stream (last kernel of step i -> first conv of step i+1) IN TIME ORDER: the window above spans the
    # warm-up steps, the barrier + device synchronisation in front of the timed region, the timed steps and the drain, so the few large
    # gaps sit at those seams; the steady state is the run of consecutive boundaries in between
    bnd = [(rs[i + 1][1] - rs[i][2], rs[i][2] - t0) for i in range(len(rs) - 1) if "inter_so3conv_c1" in rs[i + 1][0]]
    print("step boundaries of the busiest stream in time order (ms after the window start: gap in us):")
    print("   " + "  ".join(f"{at / 1e6:.0f}: {g / 1e3:.0f}" for g, at in bnd))
    best, cur = [], []
    for g, at in bnd:
        if g < 2e6:
            cur.append(g)
            if len(cur) > len(best):
                best = list(cur)
        else:
            cur = []
    if best:
        b2 = sorted(best)
        print(f"steady state = the longest run of boundaries without a seam: {len(best)} boundaries, gap median {b2[len(b2) // 2] / 1e3:.0f} us, "
              f"max {b2[-1] / 1e3:.0f} us")


if __name__ == "__main__":
    main(sys.argv[1])
