#!/bin/bash
# Kernel traces of the reference model at the launcher's default flags (fp32), of the Keras-style MNIST CNN (fp32 and
# bf16) and of the CIFAR ResNet-20 example (batch norm, pooling; fp32 = the launcher's default, and bf16) on one GPU:
# lists every kernel that ran and flags vendor BLAS / convolution / normalisation / pooling library kernels.
# usage (on the GPU box): bash tools/trace_no_vendor.sh
set -e
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DTF_SYNTH_MNIST_TRAIN=6000 DTF_SYNTH_MNIST_TEST=1000 DTF_SYNTH_CIFAR_TRAIN=2560 DTF_SYNTH_CIFAR_TEST=512
out=$root/gpurun_out/novendor
mkdir -p $out
run() {  # tag, script, flags...
  tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -o t -- python3 "$@" > $out/$tag.log 2>&1
}
run mlp_fp32 $root/initializer.py -m d -ds keras -n 1 -b 100 --device gpu
run mlp_bf16 $root/initializer.py -m d -ds keras -n 1 -b 100 --device gpu --dtype bf16
run cnn_fp32 $root/examples/mnist_cnn.py -m d -ds keras -n 1 -b 100 --device gpu
run cnn_bf16 $root/examples/mnist_cnn.py -m d -ds keras -n 1 -b 100 --device gpu --dtype bf16
run mlp_ps_fp32 $root/initializer.py -m c -cs sync -n 1 -b 100 --device gpu --transport collective
run r20_fp32 $root/examples/cifar_resnet20.py -m d -ds keras -n 1 -b 128 --device gpu
run r20_bf16 $root/examples/cifar_resnet20.py -m d -ds keras -n 1 -b 128 --device gpu --dtype bf16
python3 - $out <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
# vendor libraries, and PyTorch's own (ATen) compute kernels for the layers this framework has kernels for
vendor = re.compile(r"Cijk_|igemm|miopen|ck::|rocblas|hipblaslt|SubTensorOp|naive_conv|gemm_kernel_name|"
                    r"at::native\S*(batch_norm|max_pool|avg_pool|adaptive_avg|embedding|im2col|col2im)", re.I)
lines = []
for tag in ("mlp_fp32", "mlp_bf16", "cnn_fp32", "cnn_bf16", "mlp_ps_fp32", "r20_fp32", "r20_bf16"):
    f = glob.glob(os.path.join(out, tag, "**", "*kernel_stats.csv"), recursive=True)
    if not f:
        lines.append(f"== {tag}: no trace (see {tag}.log)")
        continue
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    bad = [r for r in rows if vendor.search(r["Name"])]
    lines.append(f"== {tag}: {len(rows)} distinct kernels, {tot/1e6:.1f} ms of kernel time, vendor-library kernels: {len(bad)}")
    for r in bad:
        lines.append(f"   VENDOR {r['Name'][:120]}  calls {r['Calls']}")
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        lines.append(f"   {100*float(r['TotalDurationNs'])/tot:5.1f}%  {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
