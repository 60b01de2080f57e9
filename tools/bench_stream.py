"""HBM streaming rates by access pattern: what the memory-bound kernels of the library can expect.

usage: python tools/bench_stream.py          (writes gpurun_out/stream_rates.txt)
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import _lib  # noqa: E402

DEV = torch.device("cuda:0")
NAMES = {0: "grid-stride, 1 x 16 B per iteration", 1: "grid-stride, 4 loads in flight", 2: "  + non-temporal stores",
         3: "  + non-temporal loads and stores"}


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    lib = _lib.kernels()
    nbytes = 1 << 31                      # 2 GiB each way: far beyond the 256 MiB last-level cache
    src = torch.empty(nbytes, dtype=torch.uint8, device=DEV).random_()
    dst = torch.empty_like(src)
    lines = []

    def run(variant, blocks, label):
        t = timeit(lambda: lib.dtf_stream_copy(_lib.ptr(src), _lib.ptr(dst), ctypes.c_longlong(nbytes), variant,
                                               blocks, _lib.stream_ptr(DEV)))
        lines.append(f"copy  {label:58s}: {2 * nbytes / t / 1e12:5.2f} TB/s")
        print(lines[-1], flush=True)

    for variant in range(4):
        for blocks in (1024, 2048, 8192):
            run(variant, blocks, f"{NAMES[variant]}, {blocks} blocks")
    for lg in range(4):
        for nt in range(4):
            run(100 + 10 * lg + nt, 0, f"one pass, {1 << lg} x 16 B per thread, non-temporal "
                                       f"{('none', 'stores', 'loads', 'loads+stores')[nt]}")
    run(200, 0, "one pass, 1 x 16 B per thread, one contiguous eighth per XCD")
    t = timeit(lambda: dst.copy_(src))
    lines.append(f"copy  torch tensor.copy_ {'':35s}: {2 * nbytes / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: dst.zero_())
    lines.append(f"fill  torch tensor.zero_ {'':35s}: {nbytes / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: src.view(torch.int32).sum())
    lines.append(f"read  torch int32 sum {'':38s}: {nbytes / t / 1e12:5.2f} TB/s")
    print("\n".join(lines[-3:]))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/stream_rates.txt", "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
