"""The three layouts of the 256x256 GEMM on one BERT-base product (65536 tokens, 3072 x 768), a few launches each,
for `bash tools/pmc_script.sh <tag> tools/pmc_gemm256_layouts.py` (counters per kernel: the template arguments in the
kernel name tell the layouts apart: <0, 0> NT, <0, 1> NN, <1, 1> TN)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm  # noqa: E402

DEV = torch.device("cuda:0")
T, N, K = 65536, 3072, 768
a = torch.randn(T, K, device=DEV).bfloat16()
b = torch.randn(N, K, device=DEV).bfloat16()
bt = b.t().contiguous()
gdy = torch.randn(T, N, device=DEV).bfloat16()
sp = gemm._big_splits(N, K, T)
for _ in range(3):
    gemm.gemm256_nt(a, b)
    gemm.gemm256(gemm.NN, a, bt, T, N, K)
    gemm.gemm256(gemm.TN, gdy, a, N, K, T, splits=sp)
torch.cuda.synchronize()
