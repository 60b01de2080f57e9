"""Times forward / data-gradient / weight-gradient of single convolutions, own kernel vs vendor library.

usage: python tools/bench_conv.py [batch] [--library] [--3x3]   (ResNet-50 strided and 3x3 shapes, batch 256 by
default; the framework's own kernels, plus the vendor library's for comparison with --library; --3x3: only the
stride-1 3x3 layers from 128 channels up, e.g. under a counter run; TF = TFLOP/s)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import conv as C  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
    shapes = [  # N, CI, H, W, CO, k, stride, pad
        (256, 256, 56, 56, 512, 1, 2, 0), (256, 512, 28, 28, 1024, 1, 2, 0), (256, 1024, 14, 14, 2048, 1, 2, 0),
        (256, 128, 56, 56, 128, 3, 2, 1), (256, 256, 28, 28, 256, 3, 2, 1), (256, 512, 14, 14, 512, 3, 2, 1),
        (256, 64, 56, 56, 64, 3, 1, 1), (256, 128, 28, 28, 128, 3, 1, 1), (256, 256, 14, 14, 256, 3, 1, 1),
        (256, 512, 7, 7, 512, 3, 1, 1),
    ]
    shapes = [(nb,) + t[1:] for t in shapes]
    if "--3x3" in sys.argv:
        shapes = [t for t in shapes if t[5] == 3 and t[6] == 1 and t[1] >= (64 if '--64' in sys.argv else 128)]
    lines = [f"{'shape (N, CI, H, W, CO, k, stride, pad)':44s} {'backend':8s} {'fwd_us':>8s} {'dgrad_us':>9s} {'wgrad_us':>9s}"
             f" {'fwd_TF':>7s} {'dgrad_TF':>8s} {'wgrad_TF':>8s}"]
    print(lines[0])
    backends = ("gemm", "library") if "--library" in sys.argv else ("gemm",)
    for (N, CI, H, W, CO, k, s, p) in shapes:
        x = torch.randn(N, CI, H, W, device=DEV).bfloat16().to(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(CO, CI, k, k, device=DEV) * 0.05).bfloat16().to(memory_format=torch.channels_last).requires_grad_(True)
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        flop = 2.0 * N * OH * OW * CO * CI * k * k
        for backend in backends:
            # "library": the vendor convolution library, called directly (the framework itself has no route to it)
            conv = C.conv2d if backend == "gemm" else \
                (lambda a, b, stride, padding: torch.nn.functional.conv2d(a, b, None, stride, padding))
            xd, wd = x.detach(), w.detach()
            yd = conv(x, wd, stride=s, padding=p)      # only the data gradient is needed
            yw = conv(xd, w, stride=s, padding=p)      # only the weight gradient is needed
            gy = torch.randn_like(yd)
            with torch.no_grad():
                tf = timeit(lambda: conv(xd, wd, stride=s, padding=p))
            td = timeit(lambda: torch.autograd.grad(yd, x, gy, retain_graph=True))
            tw = timeit(lambda: torch.autograd.grad(yw, w, gy, retain_graph=True))
            lines.append(f"{str((N, CI, H, W, CO, k, s, p)):44s} {backend:8s} {tf:8.1f} {td:9.1f} {tw:9.1f}"
                         f" {flop / tf / 1e6:7.0f} {flop / td / 1e6:8.0f} {flop / tw / 1e6:8.0f}")
            print(lines[-1], flush=True)
        del x, w, yd, yw, gy
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/conv_rates_b{nb}.txt", "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
