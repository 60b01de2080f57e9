#!/bin/bash
# Convergence evidence that can FAIL: every strategy variant on the noisy-teacher task (data/sources.py
# `_teacher_classification`: label noise in both splits, ceiling ~0.95, the reference MLP plateaus at 0.85-0.88), at 4 and
# 8 workers, 3 epochs, batch 64 per worker, Adam(1e-3), CPU plumbing mode (gloo / TCP). The last rows are a
# deliberately broken configuration (learning rate x 10): the check "within 1 point of all-reduce" must reject them.
# REFRULE rows: the reference's own rules, kept as options -- every asynchronous gradient at the full learning rate
# (reference server.py:98-102; the default here is lr / (1 + staleness)), and sync_seq = one apply per worker and round
# with a private Adam per worker (server.py:52-55,91-93).
#   usage: bash tools/convergence_teacher.sh [out_dir]
out=${1:-gpurun_out/convergence_v2}
mkdir -p $out
export DTF_SYNTH_MNIST_KIND=teacher DTF_SYNTH_MNIST_TRAIN=60000 DTF_SYNTH_MNIST_TEST=10000 OMP_NUM_THREADS=1
port=6400
run() { name=$1; n=$2; shift 2; port=$((port+1)); timeout 900 python initializer.py "$@" -n $n -b 64 --epochs 3 --device cpu --comm-timeout-s 120 --results $out/${name}_n$n.json > $out/${name}_n$n.log 2>&1 || echo "FAILED ${name}_n$n"; }
for n in 4 8; do
  run keras   $n -m d -ds keras  --loss-reduction mean
  run custom  $n -m d -ds custom --algorithm direct --loss-reduction mean
  run graph   $n -m d -ds graph -d 3 --loss-reduction mean
  run sync_c  $n -m c -cs sync  --transport collective --loss-reduction mean
  run async_c $n -m c -cs async --transport collective --loss-reduction mean
  run sync_t  $n -m c -cs sync  --transport tcp --port 0 --loss-reduction mean
  run async_t $n -m c -cs async --transport tcp --port 0 --loss-reduction mean
  DTF_ASYNC_STALENESS_LR=0 run REFRULE_async_full_rate $n -m c -cs async --transport collective --loss-reduction mean
  run sync_seq $n -m c -cs sync --transport tcp --port 0 --apply sequential --loss-reduction mean
  run BROKEN_async_lr_x10 $n -m c -cs async --transport collective --loss-reduction mean --learning-rate 0.01
  run BROKEN_keras_lr_x10 $n -m d -ds keras --loss-reduction mean --learning-rate 0.01
done
python - "$out" <<'PY'
import json, sys, glob, os
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(out + "/*.json")):
    r = json.load(open(f))
    name = os.path.basename(f)[:-5]
    acc = stale = None
    for k, v in r.items():
        if isinstance(v, dict):
            if acc is None and v.get("test_accuracy") is not None:
                acc = v["test_accuracy"]
            if v.get("mean_staleness") is not None and (stale is None or k in ("server", "rank0")):
                stale = (v.get("mean_staleness"), v.get("max_staleness"))
    rows[name] = (acc, stale)
print("%-28s %9s %12s %22s  %s" % ("run", "test_acc", "vs all-reduce", "staleness mean / max", "within 1 pt"))
for n in (4, 8):
    base = rows.get(f"keras_n{n}", (None, None))[0]
    for name, (acc, stale) in rows.items():
        if not name.endswith(f"_n{n}"):
            continue
        d = (acc - base) if (acc is not None and base is not None) else float("nan")
        st = "%.2f / %s" % (stale[0], stale[1]) if stale and stale[0] is not None else "-"
        print("%-28s %9.4f %+12.4f %22s  %s" % (name, acc if acc is not None else float("nan"), d, st,
                                                "yes" if abs(d) <= 0.01 else "NO"))
PY
