"""Single-launch all-reduce against the five-launch direct schedule with VIRTUAL ranks on one GPU.

W buffers on one device stand for the W ranks. ``fused``: the W ranks are the rows of ONE grid (resident together by
construction, so the in-kernel barriers cost what they cost on hardware queues that really run side by side -- several
processes sharing one GPU are time-sliced by the queue scheduler and say nothing about it). ``direct``: rank after rank
on one stream (a valid barrier), 2 kernels per rank and phase. Both move the same bytes; the table gives the time of
the whole W-rank collective and the device memory traffic it implies.

    python tools/bench_fused_virtual.py [--worlds 2,4,8] [--sizes-mib 1,16,48] [--dtype bf16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--sizes-mib", default="1,16,48")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from distributed_tensorflow_amd.comm.ipc import LocalPeers
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    print(f"{'world':>5} {'MiB':>6} {'direct_us':>10} {'fused_us':>10} {'fused/direct':>12} {'fused TB/s':>10} status")
    for W in [int(v) for v in a.worlds.split(",")]:
        for mib in [float(v) for v in a.sizes_mib.split(",")]:
            n = int(mib * 2 ** 20) // dt.itemsize // (W * 64) * (W * 64)
            bufs = [((torch.arange(n, device=dev) % 7) + r).to(dt) for r in range(W)]
            want = sum(((torch.arange(n, device=dev) % 7) + r).float() for r in range(W)).to(dt)
            peers = LocalPeers(bufs)
            res = {}
            for alg in ("direct", "fused"):
                for r, b in enumerate(bufs):
                    b.copy_(((torch.arange(n, device=dev) % 7) + r).to(dt))
                peers.all_reduce(alg)
                torch.cuda.synchronize()
                ok = all(torch.equal(b, want) for b in bufs)
                for _ in range(3):
                    peers.all_reduce(alg)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    peers.all_reduce(alg)
                e1.record()
                torch.cuda.synchronize()
                res[alg] = (e0.elapsed_time(e1) * 1e3 / a.iters, ok)
            moved = W * (W + 1 + 2 * (W - 1)) / W * n * dt.itemsize      # per rank: read W shards, write 1, pull W-1 (r+w)
            st = peers.fused_status()
            print(f"{W:5d} {mib:6.1f} {res['direct'][0]:10.1f} {res['fused'][0]:10.1f} "
                  f"{res['fused'][0] / res['direct'][0]:12.2f} {moved / res['fused'][0] / 1e6:10.2f} "
                  f"{'ok' if res['direct'][1] and res['fused'][1] and not any(st) else 'WRONG ' + str(st)}", flush=True)


if __name__ == "__main__":
    main()
