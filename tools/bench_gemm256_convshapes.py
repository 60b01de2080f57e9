"""Upper bound for a gathered-operand form of the 256x256 LDS-DMA kernel: the plain products with the dimensions of
the 3x3 convolutions of ResNet-50's 256- and 512-channel stages at batch 1024 (im2col already materialised, so the
operand is read 9 times from memory instead of from the cache), on the 256x256 kernel, the 128x128 kernel and the
vendor library.  usage: python tools/bench_gemm256_convshapes.py [MxNxK ...]   (default: the three 3x3 shapes; the 1x1
products of the 256- to 2048-channel stages are e.g. 200704x1024x256 200704x256x1024 50176x2048x512 50176x512x2048)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm  # noqa: E402
from tools.bench_kernels import DEV, timeit  # noqa: E402


def main():
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]] or [(200704, 256, 2304), (50176, 512, 4608),
                                                                            (802816, 128, 1152)]
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device=DEV).bfloat16()
        b = torch.randn(N, K, device=DEV).bfloat16()
        bt = b.t().contiguous()
        gdy = torch.randn(M, N, device=DEV).bfloat16()
        r = {}
        for rnd in range(3):
            for name, fn in (("nt256", lambda: gemm.gemm256_nt(a, b)),
                             ("nt128", lambda: gemm._launch(gemm.NT, a, b, M, N, K, K, K, None, None, torch.bfloat16, None)),
                             ("ntlib", lambda: torch.matmul(a, b.t())),
                             ("nn256", lambda: gemm.gemm256(gemm.NN, a, bt, M, N, K)),
                             ("tn256", lambda: gemm.gemm256(gemm.TN, gdy, a, N, K, M, splits=max(1, 512 // (((N + 255) // 256) * ((K + 255) // 256))))),
                             ("tnlib", lambda: torch.matmul(gdy.t(), a))):
                t = timeit(fn, iters=5, warmup=2) * 1e6
                r[name] = min(r.get(name, 1e30), t)
        fl = 2.0 * M * N * K / 1e6
        print(f"{M}x{N}x{K}: " + " ".join(f"{k}={v:.0f}us/{fl / v:.0f}TF" for k, v in r.items()), flush=True)
        del a, b, bt, gdy


if __name__ == "__main__":
    main()
