"""Micro-benchmarks of the HIP kernels against their roofline and the library equivalents.

Usage (on a GPU box):  python tools/bench_kernels.py [--out gpurun_out/kernels.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import bucket as B  # noqa: E402
from distributed_tensorflow_amd.ops import gemm  # noqa: E402
from distributed_tensorflow_amd.ops import optim as K  # noqa: E402
from distributed_tensorflow_amd.ops.softmax_ce import softmax_cross_entropy  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=20, warmup=5):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    res = {}

    # ---- optimizer kernels: bytes = 16 read + 12 written (+2 shadow) per element
    for n in (407050 // 64 * 64 + 64, 25_600_000, 110_000_000 // 64 * 64):
        p = torch.randn(n, device=DEV); g = torch.randn(n, device=DEV)
        m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
        sh = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
        h = K.OptHyper(DEV)
        t = timeit(lambda: K.adam_apply(p, g, m, v, h, shadow=sh))
        res[f"adam_fp32grad_shadow_n{n}"] = {"ms": t * 1e3, "TBps": n * 30 / t / 1e12}
        gb = g.to(torch.bfloat16)
        t = timeit(lambda: K.adam_apply(p, gb, m, v, h))
        res[f"adam_bf16grad_n{n}"] = {"ms": t * 1e3, "TBps": n * 26 / t / 1e12}
        t = timeit(lambda: K.sgdm_apply(p, g, m, h, shadow=sh))
        res[f"sgdm_fp32grad_shadow_n{n}"] = {"ms": t * 1e3, "TBps": n * 22 / t / 1e12}
        ref_opt = torch.optim.Adam([p.requires_grad_(True)], fused=True)
        p.grad = g
        t = timeit(lambda: ref_opt.step())
        res[f"torch_fused_adam_n{n}"] = {"ms": t * 1e3, "TBps": n * 28 / t / 1e12}
        p.requires_grad_(False)
        del p, g, m, v, sh, gb, ref_opt

    # ---- bucket flatten: ResNet-50-like tensor list
    sizes = [64 * 3 * 49] + [256 * 64, 64 * 64 * 9, 64 * 256] * 3 + [512 * 128, 128 * 128 * 9, 128 * 512] * 4 + \
            [1024 * 256, 256 * 256 * 9, 256 * 1024] * 6 + [2048 * 512, 512 * 512 * 9, 512 * 2048] * 3 + [2048 * 1000]
    ts = [torch.randn(s, device=DEV, dtype=torch.bfloat16) for s in sizes]
    offs, o = [], 0
    for t_ in ts:
        offs.append(o); o += (t_.numel() + 63) // 64 * 64
    flat = torch.empty(o, device=DEV, dtype=torch.float32)
    t = timeit(lambda: B.flatten(ts, flat, offs, scale=0.125))
    res["flatten_resnet50_bf16_to_fp32"] = {"ms": t * 1e3, "TBps": o * 6 / t / 1e12, "elements": o}
    t = timeit(lambda: torch.cat([x.float().mul_(0.125) for x in ts]))
    res["torch_cat_cast_scale_equivalent"] = {"ms": t * 1e3}

    # ---- softmax CE
    for (Bn, C, dt) in [(256, 1000, torch.bfloat16), (16384, 30522, torch.bfloat16), (65536, 10, torch.float32)]:
        x = torch.randn(Bn, C, device=DEV).to(dt).requires_grad_(True)
        y = torch.randint(0, C, (Bn,), device=DEV)

        def f():
            x.grad = None
            softmax_cross_entropy(x, y, "mean").backward()

        def fr():
            x.grad = None
            torch.nn.functional.cross_entropy(x.float(), y).backward()
        t = timeit(f, iters=10)
        tr = timeit(fr, iters=10)
        nb = Bn * C * x.element_size() * 3
        res[f"softmax_ce_fwd_bwd_{Bn}x{C}_{str(dt)[6:]}"] = {"ms": t * 1e3, "TBps": nb / t / 1e12, "torch_ms": tr * 1e3}
        del x

    # ---- GEMM vs library
    shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (65536, 768, 768), (65536, 3072, 768), (65536, 768, 3072),
              (802816, 256, 64), (802816, 64, 256), (200704, 512, 128), (50176, 1024, 256), (12544, 2048, 512),
              (256, 1000 // 8 * 8, 2048), (25600, 512, 784)]
    if a.quick:
        shapes = shapes[1:4]
    for (M, N, Kd) in shapes:
        A = torch.randn(M, Kd, device=DEV).to(torch.bfloat16)
        Bm = torch.randn(N, Kd, device=DEV).to(torch.bfloat16)
        fl = 2.0 * M * N * Kd
        t = timeit(lambda: gemm.gemm_nt(A, Bm), iters=10, warmup=3)
        tl = timeit(lambda: torch.matmul(A, Bm.t()), iters=10, warmup=3)
        r = {"nt_ms": t * 1e3, "nt_TF": fl / t / 1e12, "lib_ms": tl * 1e3, "lib_TF": fl / tl / 1e12}
        Bt = Bm.t().contiguous()
        t = timeit(lambda: gemm.gemm_nn(A, Bt), iters=10, warmup=3)
        r.update(nn_ms=t * 1e3, nn_TF=fl / t / 1e12)
        # wgrad shape: contraction over M
        G = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        t = timeit(lambda: gemm.gemm_tn(G, A), iters=10, warmup=3)
        tl = timeit(lambda: torch.matmul(G.t(), A), iters=10, warmup=3)
        r.update(tn_ms=t * 1e3, tn_TF=fl / t / 1e12, tn_lib_ms=tl * 1e3, tn_lib_TF=fl / tl / 1e12)
        res[f"gemm_{M}x{N}x{Kd}"] = r
        print(f"gemm {M}x{N}x{Kd}: " + " ".join(f"{k}={v:.1f}" for k, v in r.items() if k.endswith("TF")), flush=True)
        del A, Bm, Bt, G

    # ---- fused attention (BERT-base shape) against the unfused GEMM + softmax path
    from distributed_tensorflow_amd.ops import attention as ATT
    Bq, Sq, nh, hd = 64, 512, 12, 64
    qkv = (torch.randn(Bq, Sq, 3, nh, hd, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    do = torch.randn(Bq, Sq, nh, hd, device=DEV).to(torch.bfloat16)
    fl_f = 4.0 * Bq * nh * Sq * Sq * hd
    for name, on in (("flash", True), ("unfused", False)):
        ATT.set_flash(on)
        with torch.no_grad():
            tf_ = timeit(lambda: ATT.self_attention(qkv), iters=10, warmup=3)

        def fb():
            qkv.grad = None
            ATT.self_attention(qkv).backward(do)
        tt = timeit(fb, iters=10, warmup=3)
        res[f"attention_{name}_b64_s512_h12"] = {"fwd_ms": tf_ * 1e3, "fwd_TF": fl_f / tf_ / 1e12, "fwd_bwd_ms": tt * 1e3,
                                                 "fwd_bwd_TF_useful": 3.5 * fl_f / tt / 1e12}
    ATT.set_flash(True)
    del qkv, do

    # ---- peer-pointer collective kernels on one device (virtual ranks: "peers" are local buffers),
    #      bytes = what one rank's kernels read + write for a full all-reduce
    from distributed_tensorflow_amd.comm.ipc import LocalPeers
    for world, n, dt in [(8, 25_600_000, torch.bfloat16), (8, 25_600_000, torch.float32), (8, 65536, torch.float32)]:
        bufs = [torch.randn(n, device=DEV).to(dt) for _ in range(world)]
        lp = LocalPeers(bufs)
        esz = bufs[0].element_size()
        for alg in ("direct", "oneshot", "ring"):
            t = timeit(lambda: lp.all_reduce(alg, scale=1.0 / world), iters=5, warmup=2)
            per_rank = {"direct": n * esz * (1 + 1.0 / world) + n * esz * 2 * (world - 1) / world,
                        "oneshot": n * esz * (world + 3), "ring": n * esz * (3 * (world - 1) / world + 2 * (world - 1) / world)}[alg]
            res[f"p2p_{alg}_w{world}_n{n}_{str(dt)[6:]}"] = {"ms_all_virtual_ranks": t * 1e3,
                                                             "ms_per_rank": t * 1e3 / world,
                                                             "TBps_local": per_rank * world / t / 1e12}
        del bufs, lp

    for k, v in res.items():
        print(k, json.dumps(v))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
