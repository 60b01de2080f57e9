"""Memory-bound 1x1-convolution products in isolation: time and effective HBM rate of the expand (narrow in, wide
out) and reduce (wide in, narrow out) shapes of ResNet-50 at batch 1024, with and without the fused epilogues.

usage: python tools/bench_expand.py        (writes gpurun_out/expand_rates.txt)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm as G  # noqa: E402

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
    return s.elapsed_time(e) / iters * 1e-3


def main():
    lines = []
    only = sys.argv[1:]
    shapes = [(3211264, 64, 256), (802816, 128, 512), (200704, 256, 1024), (50176, 512, 2048),
              (3211264, 256, 64), (802816, 512, 128), (200704, 1024, 256), (3211264, 64, 64)]
    if only:
        shapes = [tuple(int(v) for v in o.split(",")) for o in only]
    for M, K, N in shapes:
        a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
        w_nn = torch.randn(K, N, device=DEV, dtype=torch.bfloat16) * 0.05
        w_nt = torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.05
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        add = torch.randn(M, N, device=DEV, dtype=torch.bfloat16)
        stats = torch.empty(G.col_stats_rows(M, N), 2, N, device=DEV, dtype=torch.float32)
        io = (M * K + M * N) * 2
        bstats = torch.rand(4, N, device=DEV, dtype=torch.float32)
        # the combinations the ResNet-50 step launches: forward = NT (+ statistics of the output for the norm that
        # follows), data gradient = NN (+ the residual branch's gradient, + the sums of the preceding norm's backward)
        cases = {
            "NT plain": (lambda: G.gemm_raw(G.NT, a, w_nt, out, M, N, K, K, K, N), io),
            "NT + column statistics": (lambda: G.gemm_raw(G.NT, a, w_nt, out, M, N, K, K, K, N, col_stats=stats), io),
            "NN plain": (lambda: G.gemm_raw(G.NN, a, w_nn, out, M, N, K, K, N, N), io),
            "NN + accumulate input": (lambda: G.gemm_raw(G.NN, a, w_nn, out, M, N, K, K, N, N, add=add), io + M * N * 2),
            "NN + norm-backward sums": (lambda: G.gemm_raw(G.NN, a, w_nn, out, M, N, K, K, N, N, col_stats=stats,
                                                            bn_grad=(add, bstats, None, 2)), io + M * N * 2),
            "NN + accumulate + sums": (lambda: G.gemm_raw(G.NN, a, w_nn, out, M, N, K, K, N, N, add=add, col_stats=stats,
                                                           bn_grad=(add, bstats, None, 2)), io + 2 * M * N * 2),
        }
        for name, (fn, nbytes) in cases.items():
            t = timeit(fn)
            lines.append(f"M {M:8d} K {K:5d} N {N:5d}  {name:24s} {t * 1e6:8.1f} us  {nbytes / t / 1e12:5.2f} TB/s"
                         f"  {2 * M * N * K / t / 1e12:6.1f} TFLOP/s")
            print(lines[-1], flush=True)
        del a, out, add
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/expand_rates.txt", "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
