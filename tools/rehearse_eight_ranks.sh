#!/bin/bash
# Rehearsal of the whole-node launch on a ONE-GPU box: eight ranks of bench.py share the GPU and talk over gloo (RCCL
# refuses several ranks on one device), small per-rank batch. It proves the launch, hooks, bucket order, side streams,
# the HIP-IPC peer mapping with eight processes and the replica check at the world size of a node; the times say
# nothing about xGMI.   usage (GPU box): bash tools/rehearse_eight_ranks.sh [strategies...]
export DTF_COMM=gloo DTF_BENCH_SHARE_GPU=1
mkdir -p gpurun_out/rehearse8
list=${@:-keras sync ipc-fused ipc-direct}
for s in $list; do
  tag=$(echo $s | tr ':' '_')
  timeout -k 10 170 python3 bench.py --gpus 8 --strategy $s --batch 16 --steps 3 --warmup 1 --phase-steps 1 --comm-timeout-s 120 > gpurun_out/rehearse8/$tag.json 2> gpurun_out/rehearse8/$tag.err
  rc=$?
  echo "$s rc=$rc $(python3 - gpurun_out/rehearse8/$tag.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("ranks_seen", d["ranks_seen"], d["comm_backend"], "ms/step", d["ms_per_step"], "agree", d["replicas_agree"], "exposed", d["exposed_comm_fraction"], "phases", d["phase_ms"])
except Exception as e:
    print("no json:", e)
PY
)"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/rehearse8/$tag.err; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then exit 1; fi; fi
done
