"""Per-launch roofline of the GEMM / convolution kernels inside one ResNet-50 training step.

usage: python tools/roofline_step.py [batch]     (default 1024; writes gpurun_out/roofline_step_b<batch>.txt)

Every launch of the 128-wide kernel family is bracketed with device events (weight gradients kept on the main
stream so that the times do not overlap), described by its dimensions and fused store-phase features, and priced
against two floors: the bytes it has to move at least (every operand read once, the result written once) at the
6.0 TB/s a one-pass device copy reaches, and its floating-point work at 1.3 PFLOP/s (what the 256x256 kernel
sustains on large plain products). The table lists, per distinct launch, calls per step, time, the higher of the
two floors and the excess over it, sorted by total excess.
"""
import os
import sys

os.environ.setdefault("DTF_WGRAD_STREAM", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd import bench_lib  # noqa: E402
from distributed_tensorflow_amd.comm.backend import Comm  # noqa: E402
from distributed_tensorflow_amd.ops import gemm as G  # noqa: E402

HBM = 6.0e12
MFMA = 1.3e15
KIND = {0: "gemm", 1: "fwd", 2: "dgrad", 3: "wgrad", 4: "wgradT"}
LAY = {0: "NT", 1: "NN", 2: "TN"}


def floors(d):
    M, N, K, b = d["M"], d["N"], d["K"], d["batch"]
    ob = 4 if d["fp32_out"] else 2
    c = d["conv"]
    if c == 0:
        flop = 2.0 * M * N * K * b
        rd = (K * (M + N) if d["layout"] == 2 else M * K + K * N) * 2 * b
        wr = M * N * ob * b * (4 if d["up2"] else 1)     # up2 also writes the three zero pixels of every 2x2 patch
    else:
        NB, H, W, OH, OW, CI, CO, KH, KW, S = (d[k] for k in ("NB", "H", "W", "OH", "OW", "CI", "CO", "KH", "KW", "S"))
        flop = 2.0 * NB * OH * OW * CO * CI * KH * KW
        px_in = NB * (OH * OW if (KH == 1 and KW == 1) else H * W)     # a strided 1x1 touches only its own pixels
        wgt = KH * KW * CI * CO * 2
        if c == 1:
            rd, wr = px_in * CI * 2 + wgt, NB * OH * OW * CO * ob
        elif c == 2:
            rd, wr = NB * OH * OW * CO * 2 + wgt, NB * H * W * CI * ob
        else:
            rd, wr = px_in * CI * 2 + NB * OH * OW * CO * 2, KH * KW * CI * CO * ob
    if d["add"]:
        rd += M * N * 2 + (M * N // 8 if d["add_mask"] else 0)
    if d["bn_grad"]:
        rd += M * N * 2 + (M * N // 8 if d["bn_mask"] else 0)
    return flop, rd + wr


def describe(d):
    if d["conv"]:
        s = (f"{KIND[d['conv']]:6s} {d['KH']}x{d['KW']}/{d['S']} {d['CI']:4d}->{d['CO']:4d} @{d['H']}x{d['W']}")
    else:
        s = f"{'gemm':6s} {LAY[d['layout']]} M{d['M']} N{d['N']} K{d['K']}" + (f" x{d['batch']}" if d["batch"] > 1 else "")
    f = [k for k in ("add", "add_mask", "stats", "bn_grad", "in_bn", "up2", "bias") if d[k]]
    if d["splits"] > 1:
        f.append(f"splitK{d['splits']}")
    if d["tile"]:
        f.append("tile64")
    return s + (" [" + ",".join(f) + "]" if f else "")


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    st = bench_lib.make_resnet50_strategy(Comm(0, 1, dev), bucket_mb=bench_lib.BUCKET_MB, grad_dtype=torch.bfloat16,
                                          strategy="keras")
    x, y = bench_lib.synthetic_imagenet_batch(batch, dev, size=224, seed=0)
    step = st.step_callable()
    for _ in range(4):
        step(x, y)
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    G.launch_log = log = []
    s0.record()
    step(x, y)
    s1.record()
    torch.cuda.synchronize()
    G.launch_log = None
    step_ms = s0.elapsed_time(s1)
    rows = {}
    for d, e0, e1 in log:
        us = e0.elapsed_time(e1) * 1e3
        flop, byts = floors(d)
        r = rows.setdefault(describe(d), [0, 0.0, flop, byts])
        r[0] += 1
        r[1] += us
    out = [f"ResNet-50 bf16, batch {batch}, one MI355X: {len(log)} launches of the 128-wide GEMM / convolution kernels in a "
           f"{step_ms:.1f} ms step (events around every launch, weight gradients on the main stream)",
           "floor = max(bytes / 6.0 TB/s, flop / 1.3 PFLOP/s); excess = time - floor, summed over the calls",
           f"{'launch':74s} {'calls':>5s} {'us/call':>8s} {'floor':>7s} {'bound':>5s} {'TB/s':>5s} {'TFLOP/s':>7s} {'excess_us':>9s}"]
    tot_t = tot_f = 0.0
    table = []
    for name, (n, us, flop, byts) in rows.items():
        t = us / n
        fm, fc = byts / HBM * 1e6, flop / MFMA * 1e6
        fl = max(fm, fc)
        table.append((n * (t - fl), name, n, t, fl, "mem" if fm >= fc else "mfma", byts / t / 1e6, flop / t / 1e6))
        tot_t += us
        tot_f += n * fl
    for ex, name, n, t, fl, bound, tbs, tfs in sorted(table, reverse=True):
        out.append(f"{name:74s} {n:5d} {t:8.1f} {fl:7.1f} {bound:>5s} {tbs:5.2f} {tfs:7.0f} {ex:9.0f}")
    out.append(f"total {tot_t / 1e3:.2f} ms in these launches, {tot_f / 1e3:.2f} ms at the floors, excess {(tot_t - tot_f) / 1e3:.2f} ms")
    text = "\n".join(out)
    print(text)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/roofline_step_b{batch}.txt", "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main()
