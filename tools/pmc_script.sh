#!/bin/bash
# Hardware counters of the kernels one script launches: three rocprofv3 --pmc passes (each a run of its own),
# summarised per kernel. usage (on the GPU box): bash tools/pmc_script.sh <tag> <script.py> [args...]
set -e
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
run() {  # name, counters... (the script and its arguments come from the enclosing call)
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o p -- \
    python3 $GRAFT_REPO_ROOT/$script "${ARGS[@]}" > $out/$name.log 2>&1
}
ARGS=("$@")
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run rd FETCH_SIZE TCC_HIT_sum
run wr WRITE_SIZE TCC_MISS_sum
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py $out/sq $out/rd $out/wr > $out/summary.txt
cat $out/summary.txt
