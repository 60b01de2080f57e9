#!/bin/bash
# One command for an 8-GPU MI355X node: the weak-scaling curve of the headline configuration, every other strategy
# at 8 ranks, BERT-base under its default schedule, and the collective micro-benchmark; writes one table.
#   usage:  bash tools/scale_sweep.sh [out_dir]          (default out_dir: gpurun_out/scale)
# Needs as many GPUs as the largest --gpus below (RCCL refuses several ranks on one device). Every run is bounded
# by `timeout`; a run that hangs or crashes ends the sweep (nothing further is started on a GPU that faulted).
set -u
out=${1:-gpurun_out/scale}
mkdir -p "$out"
ngpu=$(python3 -c 'import torch; print(torch.cuda.device_count())')
top=${DTF_SWEEP_MAX_GPUS:-8}
if [ "$ngpu" -lt "$top" ]; then echo "this node shows $ngpu GPUs, the sweep needs $top (set DTF_SWEEP_MAX_GPUS to sweep fewer)"; exit 2; fi
steps=${DTF_SWEEP_STEPS:-20}; warm=${DTF_SWEEP_WARMUP:-5}

run() {   # run <tag> <bench args...>
  tag=$1; shift
  timeout -k 10 900 python3 bench.py "$@" --steps "$steps" --warmup "$warm" > "$out/$tag.json" 2> "$out/$tag.err"
  rc=$?
  echo "$tag rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then tail -5 "$out/$tag.err"; summarise; exit 1; fi
}

summarise() {
python3 - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows = []
for p in sorted(glob.glob(os.path.join(out, "*.json"))):
    tag = os.path.basename(p)[:-5]
    try:
        d = json.loads([l for l in open(p).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        rows.append((tag, None)); continue
    rows.append((tag, d))
base = {}
for tag, d in rows:
    if d and d["n_gpus"] == 1:
        base[d["config"]["model"]] = d["value"]
with open(os.path.join(out, "table.txt"), "w") as f:
    hdr = f"{'run':28} {'model':14} {'strategy':18} {'gpus':>4} {'value':>12} {'ms/step':>9} {'eff':>6} {'exposed':>8} {'bwd_slow':>8} {'agree':>6}"
    print(hdr); f.write(hdr + "\n")
    for tag, d in rows:
        if d is None:
            line = f"{tag:28} (no JSON line)"
        else:
            m = d["config"]["model"]
            eff = d["value"] / (base[m] * d["n_gpus"]) if m in base else float("nan")
            line = (f"{tag:28} {m:14} {d['config']['strategy']:18} {d['n_gpus']:4d} {d['value']:12.1f} {d['ms_per_step']:9.2f} "
                    f"{eff:6.3f} {str(d.get('exposed_comm_fraction')):>8} {str(d.get('backward_slowdown')):>8} {str(d.get('replicas_agree')):>6}")
        print(line); f.write(line + "\n")
PY
}

for n in 1 2 4 8; do
  [ "$n" -le "$top" ] && run "resnet50_keras_n$n" --gpus "$n"
done
for s in custom:direct custom:multiring custom:ring ipc-direct ipc-fused sync async graph:3; do
  run "resnet50_$(echo $s | tr ':' '_')_n$top" --gpus "$top" --strategy "$s"
done
run "bert_n1" --model bert --gpus 1
run "bert_default_n$top" --model bert --gpus "$top"
run "bert_keras_n$top" --model bert --gpus "$top" --strategy keras
timeout -k 10 900 python3 tools/bench_collectives.py --gpus "$top" > "$out/collectives.txt" 2> "$out/collectives.err"; echo "collectives rc=$?"
summarise
