"""LayerNorm forward / backward on the BERT-base activation ([tokens, 768], residual add fused) by rows per wave of the
forward kernel.  usage: python tools/bench_layernorm.py [tokens]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import layernorm as LN, _lib  # noqa: E402
from tools.bench_kernels import DEV, timeit  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
H = 768
x = torch.randn(T, H, device=DEV).bfloat16()
r = torch.randn(T, H, device=DEV).bfloat16()
g = torch.ones(H, device=DEV)
b = torch.zeros(H, device=DEV)
lib = _lib.kernels()
lib.dtf_ln_set_fwd_rows.argtypes = [ctypes.c_int]
ref = None
for rows in (1, 2, 4, 0):
    lib.dtf_ln_set_fwd_rows(rows)
    y = LN.layer_norm(x, g, b, residual=r)
    if ref is None:
        ref = y
    assert torch.equal(y, ref)
    t = timeit(lambda: LN.layer_norm(x, g, b, residual=r), iters=10, warmup=2)
    print(f"rows per wave {rows}: {t * 1e6:7.1f} us  {3 * T * H * 2 / t / 1e12:.2f} TB/s", flush=True)
lib.dtf_ln_set_fwd_rows(0)
