#!/bin/bash
# Collects the profiles committed under profiles/ for one round (run on the GPU box through gpurun, from the repo root):
#   bash profiles/collect.sh r02
# rocprofv3 passes (kernel trace / PMC in SEPARATE runs, the program itself after `--`), summarised on the box into small text / json
# files under gpurun_out/ (the sqlite traces are deleted: gpurun_out is capped at 64 MiB).
set -u
tag=${1:-rXX}
R=$PWD
O=$R/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 "$@" > $O/$name.log 2>&1; }
# (--no-cpu-baseline --no-extras only drop legs that run AFTER the timed region -- the CPU oracle, the stage-2 / single-scan latency legs --
# whose launches of other batch sizes would mix into the per-kernel averages of the trace)
# default (overlapped) schedule and the serial schedule (every kernel on one stream: a duration is the kernel's own)
run trace_default --kernel-trace --stats -d $O/prof_default -o default -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras
grep '^{' $O/trace_default.log | tail -1 > $O/${tag}_bench_line_under_rocprof.json
db=$(find $O/prof_default -name '*_results.db' | head -1)
python3 profiles/summarize_rocpd.py $db > $O/${tag}_kernel_stats.txt
python3 profiles/timeline_rocpd.py $db > $O/${tag}_stream_timeline.txt 2>&1
echo "# --- inside two steady-state steps (profiles/scripts/gap_rocpd.py)" >> $O/${tag}_stream_timeline.txt
python3 profiles/scripts/gap_rocpd.py $db >> $O/${tag}_stream_timeline.txt 2>&1
rm -rf $O/prof_default
run trace_serial --kernel-trace --stats -d $O/prof_serial -o serial -- python3 $R/bench.py --serial --steps 16 --warmup 3 --no-cpu-baseline --no-extras
db=$(find $O/prof_serial -name '*_results.db' | head -1)
python3 profiles/summarize_rocpd.py $db > $O/${tag}_serial_kernel_stats.txt
rm -rf $O/prof_serial
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes of the same command
run pmc_fetch --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline --no-extras
run pmc_write --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write -- python3 $R/bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline --no-extras
python3 profiles/pmc_traffic.py $(find $O/pmc_fetch -name '*_results.db' | head -1) $(find $O/pmc_write -name '*_results.db' | head -1) > $O/${tag}_pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
# matrix-pipe / wave-cycle counters of the dominant kernels
run pmc_busy --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc_busy -o busy -- python3 $R/bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline --no-extras
python3 - $(find $O/pmc_busy -name '*_results.db' | head -1) > $O/${tag}_pmc_mfma.txt <<'PY'
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for name, ctr, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
    k = re.sub(r"\(.*", "", re.sub(r"^void ", "", name))
    agg[k][ctr] += val
for (name,) in c.execute("select kernel_name from counters_collection where counter_name = 'SQ_WAVE_CYCLES'"):
    cnt[re.sub(r"\(.*", "", re.sub(r"^void ", "", name))] += 1
print("# per kernel, summed over the profiled launches.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader")
print("# engines: matrix-pipe busy share of the kernel = (MFMA_BUSY / 1024) / (SQ_BUSY / 32)")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    busy = d.get("SQ_BUSY_CYCLES", 0) or 1
    print(f"{k[:70]:70s} launches {cnt[k]:4d}  matrix pipe busy {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy / 32.0:6.3f}  MFMA_BUSY {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3e}  SQ_BUSY {busy:.3e}  wave_cycles {d.get('SQ_WAVE_CYCLES', 0):.3e}")
PY
rm -rf $O/pmc_busy
# the bench lines read the newest profiles/r*_pmc_traffic.json for `roofline.traffic`: put THIS round's pass there first, so that every
# committed line of the round quotes the same counter file
cp $O/${tag}_pmc_traffic.json $R/profiles/${tag}_pmc_traffic.json
# sidecars: git blob hashes of the kernel sources these counters were collected from (bench.py marks a quoted counter `stale` when its kernel's source changed since)
python3 - $R $O/${tag}_pmc_traffic.json.sources.json $O/${tag}_pmc_mfma.txt.sources.json <<'PY'
import glob, hashlib, json, os, sys
root, outs = sys.argv[1], sys.argv[2:]
h = {}
for f in sorted(glob.glob(os.path.join(root, "etch_amd", "csrc", "*"))):
    d = open(f, "rb").read()
    h[os.path.basename(f)] = hashlib.sha1(b"blob %d\0" % len(d) + d).hexdigest()
for o in outs:
    json.dump(h, open(o, "w"), indent=1, sort_keys=True)
PY
cp $O/${tag}_pmc_traffic.json.sources.json $O/${tag}_pmc_mfma.txt $O/${tag}_pmc_mfma.txt.sources.json $R/profiles/
# the bench lines themselves (default run with the CPU baseline; forward only; the dense SMPL-X-sized config; two ranks on this one GPU)
python3 $R/bench.py > $O/${tag}_bench_line_default_run.json 2> $O/bench_default.err
python3 $R/bench.py --config 1 --steps 10 > $O/${tag}_bench_line_config1_forward_only.json 2> $O/bench_c1.err
python3 $R/bench.py --config 4 --steps 20 --warmup 3 > $O/${tag}_bench_line_config4_dense20k_smplx.json 2> $O/bench_c4.err
# (round 6) the training step: its bench line and its kernel table; the stage-1 stream's own timeline of the default run from HIP events (no tracer)
python3 $R/bench.py --train > $O/${tag}_bench_line_train.json 2> $O/bench_train.err
run trace_train --kernel-trace --stats -d $O/prof_train -o train -- python3 $R/bench.py --train --steps 6 --warmup 2 --no-cpu-baseline
python3 profiles/summarize_rocpd.py $(find $O/prof_train -name '*_results.db' | head -1) > $O/${tag}_train_kernel_stats.txt
rm -rf $O/prof_train
ETCH_PIPE_TIMING=1 python3 $R/bench.py --steps 60 --no-cpu-baseline --no-extras > /dev/null 2> $O/gaps.err
{ echo "# ETCH_PIPE_TIMING=1 python3 bench.py --steps 60 --no-cpu-baseline --no-extras: the stage-1 stream's timeline from HIP events (etch_amd/pipeline.py gap_report), no tracer attached";
  grep -v amdgpu.ids $O/gaps.err; } > $O/${tag}_boundary_gaps.txt
# SQ counters of the inter conv alone (profiles/scripts/pmc_inter_y.sh: four separate passes over profiles/scripts/time_inter_kq.py)
bash profiles/scripts/pmc_inter_y.sh $O/pmc_y > $O/${tag}_inter_conv_counters.txt 2>&1
ls -la $O
