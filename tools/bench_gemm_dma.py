"""Register-staged (one LDS buffer) against LDS-DMA (two buffers) main loop of the 128x128 GEMM kernel, per layout.

usage: python tools/bench_gemm_dma.py        (writes gpurun_out/gemm_dma_ab.txt)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import _lib  # noqa: E402
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
    lib = _lib.kernels()
    lines = []
    for M, N, K in [(8192, 4096, 4096), (200704, 256, 1024), (200704, 1024, 256), (50176, 512, 2048), (802816, 128, 512)]:
        a = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
        w_nt = torch.randn(N, K, device=DEV, dtype=torch.bfloat16) * 0.05
        w_nn = torch.randn(K, N, device=DEV, dtype=torch.bfloat16) * 0.05
        gy = torch.randn(M, N, device=DEV, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        gw = torch.empty(N, K, device=DEV, dtype=torch.bfloat16)
        cases = {
            "NT": lambda: G.gemm_raw(G.NT, a, w_nt, out, M, N, K, K, K, N),
            "NN": lambda: G.gemm_raw(G.NN, a, w_nn, out, M, N, K, K, N, N),
            "TN": lambda: G.gemm_raw(G.TN, gy, a, gw, N, K, M, N, K, K),
        }
        for name, fn in cases.items():
            ts = []
            for st in (1, 3):
                lib.dtf_gemm_set_stages(st)
                ts.append(timeit(fn))
            lib.dtf_gemm_set_stages(0)
            fl = 2.0 * M * N * K
            lines.append(f"M {M:7d} N {N:5d} K {K:5d} {name}: registers {ts[0] * 1e6:8.1f} us {fl / ts[0] / 1e12:6.0f} TFLOP/s"
                         f"   LDS-DMA {ts[1] * 1e6:8.1f} us {fl / ts[1] / 1e12:6.0f} TFLOP/s")
            print(lines[-1], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gemm_dma_ab.txt", "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
