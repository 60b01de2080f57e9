"""Per-tile wall-clock stamps of the 256x256 GEMM (loop top, operands there, k-loop done, stores issued): where a
workgroup's time goes, tile after tile.  usage: python tools/gemm256_stamps.py [persistent 0|1]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm, _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.kernels()
lib.dtf_gemm256_set_stamps.argtypes = [ctypes.c_void_p]
persist = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib.dtf_gemm256_set_persistent(persist)
for (M, N, K) in [(65536, 3072, 768), (65536, 768, 768), (8192, 2048, 768), (8192, 2048, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3):
        gemm.gemm256_nt(a, b)
    nblk = (M // 256) * (N // 256)
    st = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    lib.dtf_gemm256_set_stamps(st.data_ptr())
    gemm.gemm256_nt(a, b)
    torch.cuda.synchronize()
    lib.dtf_gemm256_set_stamps(None)
    s = st.view(nblk, 4).cpu().double() / 100.0          # us
    s = s - s[:, 0].min()
    print(f"{M}x{N}x{K}: {nblk} tiles, persistent={persist}, span {float(s[:, 3].max()):.1f} us")
    seg = s[:, 1:] - s[:, :-1]
    for name, col in (("top wait + barrier", 0), ("k-loop", 1), ("bias + next prologue + epilogue issue", 2)):
        print(f"  {name:40s} median {float(seg[:, col].median()):6.2f}  p10 {float(seg[:, col].kthvalue(max(1, nblk // 10)).values):6.2f}"
              f"  p90 {float(seg[:, col].kthvalue(max(1, nblk * 9 // 10)).values):6.2f}")
    if persist and nblk > 256:
        rounds = nblk // 256
        t = s.view(rounds, 256, 4)                        # tile it = r * 256 + b
        gap = t[1:, :, 0] - t[:-1, :, 3]                  # stores issued -> next loop top
        per = t[1:, :, 0] - t[:-1, :, 0]
        print(f"  tile period median {float(per.median()):.2f} us; stores issued -> next top {float(gap.median()):.2f}")
