"""Epilogue cost of the 256x256 GEMM at a BERT FFN shape: plain, bias, GELU, GELU + derivative, multiply, add."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm as G  # noqa: E402


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    M, N, K = 32768, 3072, 768
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    out, d = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    g2 = torch.randn(M, K, device=dev).bfloat16()
    w2 = (torch.randn(K, N, device=dev) * 0.05).bfloat16()
    res = {
        "nt_plain_us": timeit(lambda: G.gemm256(G.NT, a, w, M, N, K, out=out)),
        "nt_bias_us": timeit(lambda: G.gemm256(G.NT, a, w, M, N, K, bias=b, out=out)),
        "nt_bias_gelu_us": timeit(lambda: G.gemm256(G.NT, a, w, M, N, K, bias=b, activation="gelu", out=out)),
        "nt_bias_gelu_grad_us": timeit(lambda: G.gemm256(G.NT, a, w, M, N, K, bias=b, activation="gelu", out=out,
                                                          grad_out=d)),
        "nn_plain_us": timeit(lambda: G.gemm256(G.NN, g2, w2, M, N, K, out=out)),
        "nn_mul_us": timeit(lambda: G.gemm256(G.NN, g2, w2, M, N, K, out=out, mul=d)),
        "nn_add_us": timeit(lambda: G.gemm256(G.NN, g2, w2, M, N, K, out=out, add=d)),
        "ew_gelu_us": timeit(lambda: G.gelu_fwd(out)),
        "ew_gelu_bwd_us": timeit(lambda: G.gelu_bwd(out, d)),
    }
    res["shape"] = [M, N, K]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
