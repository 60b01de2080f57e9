#!/bin/bash
# Per-kernel time of one of the non-headline models: rocprofv3 kernel trace of tools/bench_models.py.
# usage (on the GPU box): bash tools/profile_model.sh <tag> <model> [steps]
set -e
tag=${1:-run}; model=${2:-bert}; steps=${3:-4}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/tools/bench_models.py $model --steps $steps --warmup 3 > $out/bench.log 2>&1
