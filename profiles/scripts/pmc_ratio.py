import sqlite3, sys, re, collections
# usage: pmc_ratio.py db1 db2 : per kernel TA busy share = TA_TA_BUSY_sum / 256 / (SQ_BUSY_CYCLES / 32)
def load(db):
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master") if r[0].startswith('counters_collection')][0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for name, ctr, val in c.execute(f"select kernel_name, counter_name, value from {t}"):
        k = re.sub(r"\(.*", "", re.sub(r"^void ", "", name)); agg[k][ctr] += val; cnt[(k, ctr)] += 1
    return agg, cnt
a, ca = load(sys.argv[1]); b, cb = load(sys.argv[2])
rows = []
for k in a:
    ta = a[k].get("TA_TA_BUSY_sum", 0); sq = b.get(k, {}).get("SQ_BUSY_CYCLES", 0)
    if sq: rows.append((sq, k, ta / 256.0 / (sq / 32.0), cb[(k, "SQ_BUSY_CYCLES")]))
for sq, k, r, n in sorted(rows, reverse=True)[:22]:
    print(f"{k[:64]:64s} launches {n:4d}  kernel cycles {sq/32:.3e}  TA busy {r:5.2f}")
