"""A/B of the 256x256 deep-pipelined NT GEMM against the 128x128 kernel and the vendor library,
interleaved in one process (min over rounds).  usage: python tools/bench_gemm256.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm  # noqa: E402
from tools.bench_kernels import DEV, timeit  # noqa: E402


def main():
    res = {}
    for (M, N, K) in [(32768, 768, 768), (32768, 3072, 768), (32768, 768, 3072), (32768, 2304, 768), (4096, 4096, 4096),
                      (8192, 8192, 8192), (8192, 1024, 4096), (25600, 512, 768)]:
        a = torch.randn(M, K, device=DEV).bfloat16()
        b = torch.randn(N, K, device=DEV).bfloat16()
        r = {}
        for rnd in range(3):
            lib_ = gemm._lib.kernels()
            lib_.dtf_gemm256_set_persistent(1)
            r["k256p_us"] = min(r.get("k256p_us", 1e30), timeit(lambda: gemm.gemm256_nt(a, b), iters=10, warmup=2) * 1e6)
            lib_.dtf_gemm256_set_persistent(0)
            for name, fn in (("k256", lambda: gemm.gemm256_nt(a, b)), ("k128", lambda: gemm._launch(gemm.NT, a, b, M, N, K, K, K, None, None, torch.bfloat16, None)),
                             ("lib", lambda: torch.matmul(a, b.t()))):
                t = timeit(fn, iters=10, warmup=2) * 1e6
                r[name + "_us"] = min(r.get(name + "_us", 1e30), t)
        bt = b.t().contiguous()                       # [K, N]
        gdy = torch.randn(M, N, device=DEV).bfloat16()  # weight-gradient shape: dW[N, K] = dY[M, N]^T @ X[M, K]
        gemm.set_big_tile(True)
        sp = gemm._big_splits(N, K, M)
        for rnd in range(3):
            gemm.set_big_tile(True)
            r["nn256_us"] = min(r.get("nn256_us", 1e30), timeit(lambda: gemm.gemm256(gemm.NN, a, bt, M, N, K), iters=10, warmup=2) * 1e6)
            if sp:
                r["tn256_us"] = min(r.get("tn256_us", 1e30), timeit(lambda: gemm.gemm256(gemm.TN, gdy, a, N, K, M, splits=sp), iters=10, warmup=2) * 1e6)
            gemm.set_big_tile(False)
            r["nn128_us"] = min(r.get("nn128_us", 1e30), timeit(lambda: gemm.gemm_nn(a, bt), iters=10, warmup=2) * 1e6)
            r["tn128_us"] = min(r.get("tn128_us", 1e30), timeit(lambda: gemm.gemm_tn(gdy, a), iters=10, warmup=2) * 1e6)
            r["tnlib_us"] = min(r.get("tnlib_us", 1e30), timeit(lambda: torch.matmul(gdy.t(), a), iters=10, warmup=2) * 1e6)
        gemm.set_big_tile(True)
        r["tn_splits"] = sp
        fl = 2.0 * M * N * K
        for name in ("k256", "k128", "lib"):
            r[name + "_TF"] = fl / r[name + "_us"] / 1e6
        res[f"{M}x{N}x{K}"] = r
        print(f"{M}x{N}x{K}: " + " ".join(f"{k}={v:.0f}" for k, v in r.items()), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/gemm256.json", "w"), indent=1)


if __name__ == "__main__":
    main()
