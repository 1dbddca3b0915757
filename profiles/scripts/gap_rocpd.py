"""Where does the stage-1 stream idle inside a pipelined step?  Takes the stream that runs the inter convs, picks the middle steps of the timed
window, lists every gap > 40 us between consecutive kernels of that stream (after -> before) and, for the largest ones, what the other streams ran meanwhile.
    python profiles/scripts/gap_rocpd.py results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
s1 = max({r[4] for r in rows if r[4] != 0}, key=lambda s: sum(r[2] - r[1] for r in rows if r[4] == s and "inter_so3conv" in r[0]))     # the pipeline's stage-1 stream (the synchronous legs run on stream 0)
c1 = [r for r in rows if r[4] == s1 and "inter_so3conv_c1" in r[0]]
a, b = c1[len(c1) // 2][1], c1[len(c1) // 2 + 2][1]          # two consecutive steps in the middle
rs = [r for r in rows if r[4] == s1 and a <= r[1] < b]
print(f"stream {s1}: two steps = {(b - a) / 1e6:.2f} ms, {len(rs)} kernels, busy {sum(r[2] - r[1] for r in rs) / 1e6:.2f} ms")
gaps = [(rs[i + 1][1] - rs[i][2], i) for i in range(len(rs) - 1)]
for g, i in gaps:
    if g > 40e3:
        print(f"  gap {g / 1e3:7.1f} us  at +{(rs[i][2] - a) / 1e6:6.2f} ms: {rs[i][0].split('(')[0][:44]} -> {rs[i + 1][0].split('(')[0][:44]}")
for g, i in sorted(gaps, reverse=True)[:3]:
    x, y = rs[i][2], rs[i + 1][1]
    print(f"=== during the {g / 1e3:.1f} us gap before {rs[i + 1][0].split('(')[0][:40]}:")
    agg = {}
    for r in rows:
        if r[4] != s1 and r[2] > x and r[1] < y:
            k = (r[4], r[0].split('(')[0][:50])
            agg[k] = agg.get(k, 0) + min(r[2], y) - max(r[1], x)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:8]:
        print(f"     stream {k[0]:2d}  {v / 1e3:9.1f} us  {k[1]}")
