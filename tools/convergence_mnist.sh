#!/bin/bash
# Convergence comparison on MNIST(-shaped) data: all-reduce vs PS-sync vs PS-async, one epoch,
# 2 workers, batch 64 (CPU plumbing mode; same model/optimizer everywhere). Writes JSON summaries.
set -e
out=${1:-gpurun_out/convergence}
mkdir -p $out
export DTF_SYNTH_MNIST_TRAIN=${DTF_SYNTH_MNIST_TRAIN:-60000} DTF_SYNTH_MNIST_TEST=${DTF_SYNTH_MNIST_TEST:-10000} OMP_NUM_THREADS=2
run() { name=$1; shift; python initializer.py "$@" -n 2 -b 64 --device cpu --results $out/$name.json > $out/$name.log 2>&1 || echo "FAILED $name"; }
run keras   -m d -ds keras  --loss-reduction mean
run custom  -m d -ds custom --algorithm ring --loss-reduction mean
run graph   -m d -ds graph -d 1 --loss-reduction mean
run sync_c  -m c -cs sync  --transport collective
run async_c -m c -cs async --transport collective
run sync_t  -m c -cs sync  --transport tcp --port 6301
run async_t -m c -cs async --transport tcp --port 6302
run sync_seq -m c -cs sync --transport tcp --port 6303 --apply sequential
python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(out + "/*.json")):
    r = json.load(open(f))
    acc = None
    for k, v in r.items():
        if isinstance(v, dict) and v.get("test_accuracy") is not None:
            acc = v["test_accuracy"]; break
    el = [v.get("elapsed") for v in r.values() if isinstance(v, dict) and v.get("elapsed")]
    rows.append((os.path.basename(f)[:-5], acc, el[0] if el else None))
print("%-10s %10s %10s" % ("run", "test_acc", "elapsed_s"))
for n, a, e in rows:
    print("%-10s %10.4f %10.2f" % (n, a if a is not None else float("nan"), e or float("nan")))
PY
