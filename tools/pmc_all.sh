#!/bin/bash
# Hardware counters of every hand-written kernel: three rocprofv3 --pmc passes (each a run of its own) over
# tools/pmc_kernels.py, summarised per kernel. usage (on the GPU box): bash tools/pmc_all.sh <tag>
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o p -- \
    python3 $GRAFT_REPO_ROOT/tools/pmc_kernels.py > $out/$name.log 2>&1
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run rd FETCH_SIZE TCC_HIT_sum
run wr WRITE_SIZE TCC_MISS_sum
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py $out/sq $out/rd $out/wr > $out/summary.txt
wc -l $out/summary.txt
