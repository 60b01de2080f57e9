#!/bin/bash
# Per-kernel time of the flagship step: rocprofv3 kernel trace + stats of a short bench run.
# usage (on the GPU box): bash tools/profile_bench.sh <tag> [bench.py args...]
set -e
tag=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp
# no first-touch child process under the profiler (its preloaded library would ride along into the child)
export DTF_BENCH_PRETOUCH=0
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 "$@" > $out/bench.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" $out/summary.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[2], "w") as f:
    f.write(f"total kernel time {tot/1e6:.2f} ms over the whole run (warm-up, timed and phase-timing steps: see calls / kernels per step)\n")
    f.write(f"{'pct':>6} {'total_ms':>10} {'calls':>7} {'avg_us':>9}  name\n")
    for r in rows[:60]:
        f.write(f"{100*float(r['TotalDurationNs'])/tot:6.2f} {float(r['TotalDurationNs'])/1e6:10.3f} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.1f}  {r['Name'][:150]}\n")
PY
cat $out/summary.txt | head -45
