#!/bin/bash
# Instruction mix and wait cycles of the kernels one script launches (one rocprofv3 --pmc pass).
# usage (on the GPU box): bash tools/pmc_insts.sh <tag> <script.py> [args...]
set -e
tag=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d $out/insts -o p -- python3 $GRAFT_REPO_ROOT/$script "$@" > $out/insts.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py $out/insts > $out/insts.txt
cat $out/insts.txt
