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
            for name, fn in (("k256", lambda: gemm.gemm256_nt(a, b)), ("k128", lambda: gemm._launch(gemm.NT, a, b, M, N, K, K, K, None, None, torch.bfloat16, None)),
                             ("lib", lambda: torch.matmul(a, b.t()))):
                t = timeit(fn, iters=10, warmup=2) * 1e6
                r[name + "_us"] = min(r.get(name + "_us", 1e30), t)
        fl = 2.0 * M * N * K
        for name in ("k256", "k128", "lib"):
            r[name + "_TF"] = fl / r[name + "_us"] / 1e6
        res[f"{M}x{N}x{K}"] = r
        print(f"{M}x{N}x{K}: " + " ".join(f"{k}={v:.0f}" for k, v in r.items()), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/gemm256.json", "w"), indent=1)


if __name__ == "__main__":
    main()
