#!/bin/bash
# Hardware counters of the fused attention kernels (BERT-base shape): two rocprofv3 --pmc passes over
# tools/bench_attention.py, each a run of its own. usage (on the GPU box): bash tools/pmc_attention.sh <tag>
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $out/sq1 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_attention.py > $out/sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out/sq2 -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_attention.py > $out/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py $out/sq1 $out/sq2 > $out/summary.txt
grep -E "^#|kernel |flash|attn_delta" $out/summary.txt | cut -c1-260
