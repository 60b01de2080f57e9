"""Times the 128x128 GEMM kernel on the 1x1-convolution shapes of a ResNet-50 step (batch 1024) as plain products:
forward (NT) with and without the batch-norm column statistics in the store phase, and the data-gradient (NN)
product of the same layer. Prints microseconds and the effective HBM rate (bytes of A + output over time).

    python tools/bench_conv1x1_shapes.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import gemm  # noqa: E402

DEV = torch.device("cuda:0")
NT, NN = gemm.NT, gemm.NN


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    B = int(os.environ.get("BATCH", "1024"))
    rows = []
    for (hw, cin, cout) in [(56, 64, 256), (56, 256, 64), (28, 128, 512), (28, 512, 128), (14, 256, 1024),
                            (14, 1024, 256), (7, 512, 2048), (7, 2048, 512)]:
        M = B * hw * hw
        x = torch.randn(M, cin, device=DEV).bfloat16()
        w = (torch.randn(cout, cin, device=DEV) * 0.05).bfloat16()
        y = torch.empty(M, cout, device=DEV, dtype=torch.bfloat16)
        st = torch.empty(gemm.col_stats_rows(M, cout), 2, cout, device=DEV)
        dx = torch.empty_like(x)
        t_plain = timed(lambda: gemm.gemm_raw(NT, x, w, y, M, cout, cin, cin, cin, cout))
        t_stats = timed(lambda: gemm.gemm_raw(NT, x, w, y, M, cout, cin, cin, cin, cout, col_stats=st))
        t_dgrad = timed(lambda: gemm.gemm_raw(NN, y, w, dx, M, cin, cout, cout, cin, cin))   # dx = dy [M, cout] . w [cout, cin]
        gb = (M * cin + M * cout) * 2 / 1e9
        rows.append({"hw": hw, "cin": cin, "cout": cout, "M": M, "fwd_us": round(t_plain, 1),
                     "fwd_stats_us": round(t_stats, 1), "dgrad_us": round(t_dgrad, 1),
                     "fwd_TBps": round(gb / t_plain * 1e3, 2), "fwd_stats_TBps": round(gb / t_stats * 1e3, 2),
                     "dgrad_TBps": round(gb / t_dgrad * 1e3, 2)})
        print(json.dumps(rows[-1]), flush=True)
        del x, w, y, st, dx


if __name__ == "__main__":
    main()
