"""Times the fused attention kernels at BERT-base shape (or --shape B S nh): forward, forward+backward."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import attention as ATT  # noqa: E402


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
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=3, default=[64, 512, 12])
    ap.add_argument("--bias", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    B, S, nh = a.shape
    dev = torch.device("cuda:0")
    qkv = (torch.randn(B, S, 3, nh, 64, device=dev) * 0.5).to(torch.bfloat16).requires_grad_(True)
    do = torch.randn(B, S, nh, 64, device=dev).to(torch.bfloat16)
    bias = torch.zeros(B, S, device=dev) if a.bias else None
    fl = 4.0 * B * nh * S * S * 64
    with torch.no_grad():
        tf = timeit(lambda: ATT.self_attention(qkv, bias))

    def fb():
        qkv.grad = None
        ATT.self_attention(qkv, bias).backward(do)
    tt = timeit(fb)
    res = {"shape": [B, S, nh, 64], "bias": a.bias, "fwd_ms": tf, "fwd_TF": fl / tf / 1e9, "bwd_ms": tt - tf,
           "bwd_TF": 2.5 * fl / (tt - tf) / 1e9, "fwd_bwd_ms": tt}
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
