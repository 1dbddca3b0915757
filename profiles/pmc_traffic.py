#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as the
TCC counter slots require):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python profiles/pmc_traffic.py gpurun_out/pmc_fetch/fetch_results.db gpurun_out/pmc_write/write_results.db > profiles/r01_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced read -> doubled.  Calibration on this workload: mhsa_attention_kernel reads its
qkv tensor once (9.6 M rows x 192 fp32 = 7.37 GB): 2 x FETCH_SIZE = 7.37 GB; WRITE_SIZE = 2.46 GB = its output exactly.
"""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>]*>)?", name)
    if m is None:
        return name[:60]
    base, targs = m.group(1), (m.group(2) or "")
    if base == "gemm_nt_kernel":
        return base
    return base + targs.replace(" ", "")


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    out = {}
    for name, n, total in c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)):
        k = short(name)
        a = out.setdefault(k, [0, 0.0])
        a[0] += n
        a[1] += total
    return out


def main(fetch_db, write_db):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        nf, tf = f.get(k, [0, 0.0])
        nw, tw = w.get(k, [0, 0.0])
        n = max(nf, nw)
        if n == 0:
            continue
        res[k] = {"launches_profiled": n, "read_bytes_per_launch": 2.0 * tf * 1024 / max(nf, 1), "write_bytes_per_launch": tw * 1024 / max(nw, 1)}
        res[k]["bytes_per_launch"] = res[k]["read_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    json.dump(res, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
