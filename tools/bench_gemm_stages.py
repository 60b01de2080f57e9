"""A/B of the GEMM main loop with one vs two LDS buffers on the ResNet-50 / BERT shapes, interleaved
in one process (min over rounds).  usage: python tools/bench_gemm_stages.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm  # noqa: E402
from tools.bench_kernels import DEV, timeit  # noqa: E402


def main():
    res = {}
    lib = gemm._lib.kernels()
    for (M, N, Kd) in [(802816, 256, 64), (802816, 64, 256), (200704, 512, 128), (200704, 128, 512), (50176, 1024, 256),
                       (50176, 256, 1024), (12544, 2048, 512), (12544, 512, 2048), (32768, 768, 768), (32768, 3072, 768),
                       (4096, 4096, 4096)]:
        A = torch.randn(M, Kd, device=DEV).to(torch.bfloat16)
        Bm = torch.randn(N, Kd, device=DEV).to(torch.bfloat16)
        G = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        r = {}
        for rnd in range(2):
            for st in (1, 2):
                lib.dtf_gemm_set_stages(st)
                for name, fn in (("nt", lambda: gemm.gemm_nt(A, Bm)), ("nn", lambda: gemm.gemm_nn(G, Bm)),
                                 ("tn", lambda: gemm.gemm_tn(G, A))):
                    t = timeit(fn, iters=10, warmup=2) * 1e6
                    key = f"{name}_s{st}_us"
                    r[key] = min(r.get(key, 1e30), t)
        lib.dtf_gemm_set_stages(0)
        res[f"gemm_stages_{M}x{N}x{Kd}"] = r
        print(f"stages {M}x{N}x{Kd}: " + " ".join(f"{k}={v:.0f}" for k, v in r.items()), flush=True)
        del A, Bm, G

    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/gemm_stages.json", "w"), indent=1)


if __name__ == "__main__":
    main()
