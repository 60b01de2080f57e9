#!/bin/bash
# Rehearsal on a one-GPU box: two ranks of bench.py share the GPU and talk over gloo (RCCL refuses two ranks on one
# device), once per strategy, small batch. Prints per strategy: exit code, ranks_seen, backend, ms per step, whether
# the replicas agree after the run, exposed communication fraction. The ipc-* strategies still run their peer-pointer
# kernels on device memory shared through HIP IPC.   usage (GPU box): bash tools/rehearse_two_ranks.sh
export DTF_COMM=gloo DTF_BENCH_SHARE_GPU=1
mkdir -p gpurun_out/rehearse
for s in keras custom:ring custom:direct custom:multiring custom:tree custom:oneshot ipc-fused ipc-direct graph:1 sync async; do
  tag=$(echo $s | tr ':' '_')
  timeout -k 10 170 python bench.py --gpus 2 --strategy $s --batch 64 --steps 3 --warmup 1 --phase-steps 1 > gpurun_out/rehearse/$tag.json 2> gpurun_out/rehearse/$tag.err
  rc=$?
  echo "$s rc=$rc $(python - gpurun_out/rehearse/$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["ranks_seen"], d["comm_backend"], d["ms_per_step"], d["replicas_agree"], d["exposed_comm_fraction"])
except Exception as e:
    print("no json:", e)
PY
)"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/rehearse/$tag.err; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then exit 1; fi; fi
done
