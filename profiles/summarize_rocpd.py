#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd/sqlite output) kernel trace into the text summary committed under profiles/.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o NAME -- python3 bench.py ...
    python profiles/summarize_rocpd.py gpurun_out/prof/NAME_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count),"
                     " max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats  ({path.split('/')[-1]}); durations in microseconds; total kernel time {tot / 1e3:.1f} us x1e0 = {tot / 1e6:.2f} ms")
    print(f"{'kernel':90s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>7s}")
    for n, calls, total, avg, mn, mx, vg, ag, sg, lds, scr in rows:
        name = n if len(n) <= 90 else n[:87] + "..."
        print(f"{name:90s} {calls:6d} {total / 1e3:11.1f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:10.2f} {100.0 * total / tot:6.2f} {vg or 0:5d} {ag or 0:5d} {sg or 0:5d} {lds or 0:7d} {scr or 0:7d}")


if __name__ == "__main__":
    main(sys.argv[1])
