"""All-reduce time and bus bandwidth by message size, one rank per GPU, for every back end of the framework:

  rccl            ``Comm.all_reduce`` (torch.distributed: RCCL on GPUs, gloo on the CPU) -- the ``keras`` strategy's path
  custom:<alg>    the schedules of ``comm/algorithms.py`` on grouped point-to-point (direct, ring, multiring, tree,
                  oneshot) -- the ``custom`` strategy
  ipc-<alg>       the hand-written kernels on HIP-IPC peer pointers (direct, fused, oneshot, ring, tree) -- GPU only
  rs+ag           reduce-scatter followed by all-gather (``Comm``): the push / pull of the synchronous parameter server

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 \\
        tools/bench_collectives.py [--sizes-mib 0.25,1,4,16,64,256] [--dtype bf16] [--iters 20] [--only rccl,ipc-fused]
    python tools/bench_collectives.py --gpus 8          (starts the ranks itself)
    python tools/bench_collectives.py --gpus 2 --device cpu --sizes-mib 0.25,1      (plumbing run over gloo)

``DTF_COMM=gloo DTF_BENCH_SHARE_GPU=1`` deals the ranks over the GPUs that exist (a rehearsal on a one-GPU box; the
times then say nothing about xGMI). Bus bandwidth = 2 (W - 1) / W x bytes / time, the usual all-reduce convention.
Every back end is checked against the exact sum before it is timed. Rank 0 prints a table and writes
``gpurun_out/collectives_w<W>.json``.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

CUSTOM = ("direct", "ring", "multiring", "tree", "oneshot")
IPC = ("direct", "fused", "oneshot", "ring", "tree")


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--device", default="gpu", choices=["gpu", "cpu"])
    ap.add_argument("--sizes-mib", default="0.25,1,4,16,64,256")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="comma-separated back ends (default: all that apply)")
    ap.add_argument("--ipc-blocks", type=int, default=0,
                    help="CU budget (blocks) of the peer-pointer kernels; 0 = their defaults (64 for the single-launch one)")
    ap.add_argument("--ipc-threads", type=int, default=256,
                    help="widest block of the single-launch all-reduce (1024 where ranks share one GPU: blocks x ranks <= 256)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        return subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                                "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
                               + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.device == "gpu":
        if os.environ.get("DTF_BENCH_SHARE_GPU") == "1":
            local %= max(1, torch.cuda.device_count())
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")

    from distributed_tensorflow_amd.comm import algorithms as alg
    from distributed_tensorflow_amd.comm.backend import Comm
    comm = Comm.from_env(dev, timeout_s=600.0) if world > 1 else Comm(0, 1, dev)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    esz = 2 if a.dtype == "bf16" else 4

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        comm.barrier()

    backends = ["rccl"] + [f"custom:{k}" for k in CUSTOM] + ["rs+ag"]
    peers = None
    if dev.type == "cuda" and world > 1:
        try:
            from distributed_tensorflow_amd.comm.ipc import PeerGroup
            peers = PeerGroup(comm, max_blocks=a.ipc_blocks, fused_threads=a.ipc_threads)
            backends += [f"ipc-{k}" for k in IPC]
        except Exception as e:      # no peer access between these devices / processes
            if rank == 0:
                print(f"peer-pointer kernels not available here: {e}", file=sys.stderr)
    if a.only:
        backends = [b for b in backends if b in a.only.split(",")]

    results = []
    for mib in [float(v) for v in a.sizes_mib.split(",")]:
        n = max(world * 64, int(mib * 2 ** 20) // esz // (world * 64) * (world * 64))
        buf = torch.empty(n, dtype=dtype, device=dev)
        scratch = torch.empty(n, dtype=dtype, device=dev)
        shard = torch.empty(n // world, dtype=dtype, device=dev)
        # small integers: every partial sum is exact in bf16, so any schedule must give the same bits
        src = ((torch.arange(n, device=dev) % 7) + rank).to(dtype)
        want = sum(((torch.arange(n, device=dev) % 7) + r).to(torch.float32) for r in range(world)).to(dtype)

        def run(name):
            if name == "rccl":
                comm.all_reduce(buf)
            elif name.startswith("custom:"):
                alg.all_reduce(comm, buf, name.split(":", 1)[1], scratch)
            elif name == "rs+ag":
                comm.reduce_scatter(shard, buf)
                comm.all_gather(buf, shard)
            else:
                peers.all_reduce(buf, name.split("-", 1)[1])

        for name in backends:
            if world == 1 and name != "rccl":
                continue
            buf.copy_(src)
            sync()
            run(name)
            sync()
            ok = bool(torch.equal(buf, want)) if world > 1 else True
            for _ in range(2):
                run(name)
            sync()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                run(name)
            sync()
            dt = (time.perf_counter() - t0) / a.iters
            t = torch.tensor([dt], dtype=torch.float64 if dev.type == "cpu" else torch.float32, device=dev)
            if world > 1:
                import torch.distributed as dist
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
            bus = 2.0 * (world - 1) / world * n * esz / dt / 1e9 if world > 1 else 0.0
            results.append({"backend": name, "mib": round(n * esz / 2 ** 20, 3), "us": round(dt * 1e6, 1),
                            "bus_gb_s": round(bus, 2), "correct": ok})
            if rank == 0:
                print(f"{name:18s} {n * esz / 2 ** 20:9.3f} MiB {dt * 1e6:10.1f} us {bus:9.2f} GB/s  "
                      f"{'ok' if ok else 'WRONG RESULT'}", flush=True)
        del buf, scratch, shard, src, want
    if peers is not None:
        peers.check()
        peers.close()
    if rank == 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"collectives_w{world}.json"), "w") as f:
            json.dump({"world": world, "device": dev.type, "backend": comm.backend if world > 1 else "none",
                       "dtype": a.dtype, "results": results}, f, indent=1)
    bad = [r for r in results if not r["correct"]]
    comm.shutdown()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
