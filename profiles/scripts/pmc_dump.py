import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
t = [x for x in tabs if x.startswith('counters_collection')][0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name, ctr, val in c.execute(f"select kernel_name, counter_name, value from {t}"):
    k = re.sub(r"\(.*", "", re.sub(r"^void ", "", name))
    agg[k][ctr].append(val)
for k, d in agg.items():
    if not any(s in k for s in sys.argv[2:]): continue
    print(k[:60], {ctr: (len(v), f"{sum(v)/len(v):.4g}") for ctr, v in d.items()})
