"""What the index stream runs around a late step boundary of the stage-1 stream: for the largest steady-state boundary gap (seams > 8 ms excluded)
lists the kernels of every other stream from 12 ms before the gap to its end.   python profiles/scripts/boundary_rocpd.py results.db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
s1 = max({r[4] for r in rows if r[4] != 0}, key=lambda s: sum(r[2] - r[1] for r in rows if r[4] == s and "inter_so3conv" in r[0]))
m = [r for r in rows if r[4] == s1]
gaps = [(m[i + 1][1] - m[i][2], i) for i in range(len(m) - 1) if "inter_so3conv_c1" in m[i + 1][0] and m[i + 1][1] - m[i][2] < 8e6]
g, i = max(gaps)
x, y = m[i][2], m[i + 1][1]
print(f"stage-1 stream {s1}: gap {g / 1e3:.0f} us after {m[i][0].split('(')[0][:30]}; t = 0 at the gap's start")
for r in rows:
    if r[4] != s1 and r[2] > x - 12e6 and r[1] < y:
        print(f"  stream {r[4]}  {(r[1] - x) / 1e3:9.0f} .. {(r[2] - x) / 1e3:9.0f} us   {r[0].split('(')[0][:60]}")
