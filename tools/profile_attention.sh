#!/bin/bash
# Per-kernel time of the fused attention kernels: rocprofv3 kernel trace of tools/bench_attention.py.
# usage (on the GPU box): bash tools/profile_attention.sh <tag> [bench_attention args]
set -e
tag=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/fa_$tag
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_attention.py "$@" > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[1] + "/summary.txt", "w") as o:
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        line = f"{k[:90]:90s} n={len(v):4d} median={v[len(v)//2]/1e3:8.1f} us min={v[0]/1e3:8.1f} us"
        print(line); o.write(line + "\n")
PY
