#!/bin/bash
# BASELINE config 2 (CIFAR-10 ResNet-20, bf16, batch 128 per worker) with several workers sharing ONE GPU: the
# asynchronous parameter server under real concurrency (owners serve pushes from all workers at once over the
# HIP-IPC pull plane; gloo carries the control traffic because RCCL refuses several ranks on one device), next to the
# synchronous server and the mirrored all-reduce. usage (on the GPU box): bash tools/config2_ranks.sh [ranks...]
set -u
cd "$(dirname "$0")/.."
export DTF_COMM=gloo DTF_SYNTH_CIFAR_TRAIN=51200 DTF_SYNTH_CIFAR_TEST=10000
out=gpurun_out/config2
mkdir -p $out
ranks=${@:-"1 4 8"}
for n in $ranks; do
  for run in "c async" "c sync" "d keras"; do
    set -- $run
    mode=$1; strat=$2
    flag=$([ $mode = c ] && echo -cs || echo -ds)
    timeout -k 10 280 python examples/cifar_resnet20.py -m $mode $flag $strat -n $n -b 128 --dtype bf16 --epochs 2 \
      --results $out/${strat}_n$n.json > $out/${strat}_n$n.log 2>&1 || { echo "$strat n=$n failed rc=$?"; tail -5 $out/${strat}_n$n.log; exit 1; }
    echo "done $strat n=$n"
  done
done
python - $out $ranks <<'PY'
import json, sys
out, ranks = sys.argv[1], [int(r) for r in sys.argv[2:]]
print(f"{'run':8s} {'ranks':>5s} {'elapsed_s':>10s} {'images/s':>10s} {'ms/step':>9s} {'mean_stale':>10s} {'max_stale':>9s} {'test_acc':>9s}")
for n in ranks:
    for s in ("async", "sync", "keras"):
        d = json.load(open(f"{out}/{s}_n{n}.json"))
        rs = [v for k, v in d.items() if k.startswith("rank") and isinstance(v, dict)]
        el = max(float(r.get("elapsed") or 0.0) for r in rs)
        steps = max(int(r.get("steps") or 0) for r in rs)
        stale = [r["mean_staleness"] for r in rs if "mean_staleness" in r]
        mx = [r["max_staleness"] for r in rs if "max_staleness" in r]
        acc = [r["test_accuracy"] for r in rs if r.get("test_accuracy") is not None]
        ips = steps * 128 * n / el if el else 0.0
        print(f"{s:8s} {n:5d} {el:10.2f} {ips:10.0f} {(el * 1e3 / steps if steps else 0):9.2f} "
              f"{(sum(stale) / len(stale) if stale else float('nan')):10.2f} {(max(mx) if mx else '-'):>9} "
              f"{(sum(acc) / len(acc) if acc else float('nan')):9.4f}")
PY
