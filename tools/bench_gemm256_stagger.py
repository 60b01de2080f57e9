"""256x256 GEMM on the BERT-base products with the first round of workgroups started in four groups a few
microseconds apart (``dtf_gemm256_set_stagger``), against all workgroups starting together.

usage: python tools/bench_gemm256_stagger.py [tokens]     (default 65536 = batch 128 x 512)
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm  # noqa: E402
from tools.bench_kernels import DEV, timeit  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    lib = gemm._lib.kernels()
    lib.dtf_gemm256_set_stagger.argtypes = [ctypes.c_int]
    res = {}
    staggers = [0]
    for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
        a = torch.randn(T, K, device=DEV).bfloat16()
        b = torch.randn(N, K, device=DEV).bfloat16()
        bt = b.t().contiguous()
        gdy = torch.randn(T, N, device=DEV).bfloat16()
        sp = gemm._big_splits(N, K, T)
        cin = torch.randn(T, N, device=DEV).bfloat16()
        bias = torch.randn(N, device=DEV)
        der = torch.empty(T, N, device=DEV, dtype=torch.bfloat16)
        cases = {"nt": lambda: gemm.gemm256_nt(a, b), "nn": lambda: gemm.gemm256(gemm.NN, a, bt, T, N, K),
                 "nt+bias": lambda: gemm.gemm256_nt(a, b, bias),
                 "nt+bias+gelu+der": lambda: gemm.gemm256(gemm.NT, a, b, T, N, K, bias, "gelu", grad_out=der),
                 "nn+add": lambda: gemm.gemm256(gemm.NN, a, bt, T, N, K, add=cin),
                 "nn+mul": lambda: gemm.gemm256(gemm.NN, a, bt, T, N, K, mul=cin)}
        if sp:
            cases["tn"] = lambda: gemm.gemm256(gemm.TN, gdy, a, N, K, T, splits=sp)
        for name, fn in cases.items():
            row = {}
            for rnd in range(2):
                for persist in (0, 1):
                    lib.dtf_gemm256_set_persistent(persist)
                    for s in staggers:
                        lib.dtf_gemm256_set_stagger(s)
                        k = f"p{persist}_s{s}"
                        row[k] = min(row.get(k, 1e30), timeit(fn, iters=10, warmup=2) * 1e6)
            lib.dtf_gemm256_set_persistent(0)
            lib.dtf_gemm256_set_stagger(0)
            fl = 2.0 * T * N * K
            res[f"{name} {T}x{N}x{K}"] = row
            print(f"{name} {T}x{N}x{K} (base {fl / row['p0_s0'] / 1e6:.0f} TFLOP/s): "
                  + " ".join(f"{k}={v:.0f}" for k, v in row.items()), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/gemm256_stagger.json", "w"), indent=1)


if __name__ == "__main__":
    main()
