"""Headline benchmark: synchronous data-parallel training throughput, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default configuration (the headline one): ResNet-50, synthetic 224x224 images, bf16 compute,
decentralized keras-style mirrored all-reduce over RCCL. Weak scaling: the per-GPU batch is fixed, the
global batch grows with N. A step is forward + backward + bucketed gradient exchange + optimizer update;
nothing is skipped inside the timed region.

``--strategy`` runs the same step under any other strategy (``custom:ring``, ``custom:direct``, ``ipc-direct``,
``graph:2``, ``sync``, ``async``); ``--model bert`` is BERT-base MLM at sequence length 512 (sequences/sec, by
default under ``custom:direct``: the "custom bucketed all-reduce overlapped with backward" configuration on the
schedule that fits a fully connected xGMI node -- one hop, all seven links of a GPU busy, three host-issued
operations per bucket whatever the world size, where a single ring is bound by ONE link and costs 3 (W - 1)).

Rank 0 prints one JSON line. Besides the throughput it carries evidence of what ran: ``ranks_seen`` (an
all-reduce of ones over the process group), the communication backend, the bucket plan, a replica checksum
comparison after the timed loop, and -- from a few extra, untimed steps bracketed with device events --
per-phase device time with the exposed communication wait as a fraction of the step. With more than one rank the
same phases are first measured on every rank ALONE (a world-1 strategy on the same model and batch, before the
process group carries any traffic): ``backward_slowdown`` = backward at N ranks / backward alone is what the
collectives cost the backward pass they overlap with (CU and HBM contention, a transfer that blocks where it is
started), which the wait phases cannot show; ``comm_in_hook`` is the part of it spent starting collectives.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=None,
                    help="per-GPU batch (weak scaling: fixed per GPU); default 2048 images (ResNet-50: 1.5-2 %% above 1024 and\n"
                         "6 %% above 512 on one MI355X -- the per-layer reduction kernels and ~550 launch boundaries are amortised\n"
                         "over more work; 86 GiB of the 288 GB) / 384 sequences (BERT: 4-5 %% above 128, 18 %% above 64 -- every\n"
                         "workgroup of the persistent 256 x 256 GEMM then walks 9 or more output tiles; 60 GiB)")
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--bucket-mb", type=int, default=None, help="gradient bucket size (default 48 MiB)")
    ap.add_argument("--device", default="gpu", choices=["gpu", "cpu"], help="cpu is for plumbing tests only")
    ap.add_argument("--model", default="resnet50", choices=["resnet50", "bert", "bert-tiny"])
    ap.add_argument("--strategy", default=None,
                    help="keras | custom[:direct|ring|multiring|tree|oneshot] | ipc[-direct|-fused|-oneshot|-ring|-tree] | "
                         "graph[:degree] | sync | async (default: keras for resnet50, custom:direct for bert)")
    ap.add_argument("--grad-dtype", default="bf16", choices=["bf16", "fp32"], help="dtype the gradients are exchanged in")
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one hipGraph")
    ap.add_argument("--input", default="device", choices=["device", "host"],
                    help="device: one synthetic batch resident on the GPU (tf_cnn_benchmarks-style synthetic data); host: "
                         "uint8 images in host memory, gathered per step by the native loader threads, uploaded on a copy "
                         "stream one step ahead and normalised on the GPU (ResNet-50 only)")
    ap.add_argument("--phase-steps", type=int, default=3, help="untimed steps after the timed loop that measure per-phase device time (0 = skip)")
    ap.add_argument("--ref-steps", type=int, default=3,
                    help="with several ranks: steps of the single-rank reference measurement of the phases (0 = skip)")
    ap.add_argument("--comm-timeout-s", type=float, default=1200.0,
                    help="every collective of the process group fails after this many seconds instead of blocking")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: start one rank per GPU as child processes (nothing here has
        # touched the GPU yet) and pass their exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        return subprocess.call(cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # One throwaway child process writes the device's free memory once, before this process touches the GPU: the first
    # process to work through the HBM of a freshly booted box runs the step 1-1.5 % slower for as long as it lives
    # (utils/vram.py). Reported in the JSON line; DTF_BENCH_PRETOUCH=0 skips it. Not with several ranks on one GPU.
    pretouch = None
    if (a.device == "gpu" and os.environ.get("DTF_BENCH_PRETOUCH", "1") != "0"
            and os.environ.get("DTF_BENCH_SHARE_GPU") != "1"
            and "rocprof" not in os.environ.get("LD_PRELOAD", "")):      # (not under a profiler's preloaded library)
        from distributed_tensorflow_amd.utils.vram import pretouch as _pretouch
        pretouch = _pretouch(local_rank)
    if a.device == "gpu":
        if not torch.cuda.is_available():
            print("bench.py needs a GPU (use --device cpu only for plumbing tests)", file=sys.stderr)
            return 2
        # DTF_BENCH_SHARE_GPU=1 (rehearsals on a box with fewer GPUs than ranks; use with DTF_COMM=gloo, RCCL
        # refuses two ranks on one device): ranks are dealt round-robin over the GPUs that exist
        if os.environ.get("DTF_BENCH_SHARE_GPU") == "1":
            local_rank %= max(1, torch.cuda.device_count())
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")

    from distributed_tensorflow_amd import bench_lib
    from distributed_tensorflow_amd.comm.backend import Comm
    from distributed_tensorflow_amd.obs.trace import PhaseTimer
    from distributed_tensorflow_amd.ops import _lib

    if dev.type == "cuda":
        _lib.kernels()
    comm = Comm.from_env(dev, timeout_s=a.comm_timeout_s) if world > 1 else Comm(0, 1, dev)
    bucket_mb = a.bucket_mb or bench_lib.BUCKET_MB
    gdt = torch.bfloat16 if a.grad_dtype == "bf16" else torch.float32
    if a.model == "resnet50":
        strategy = a.strategy or "keras"
        batch = a.batch or int(os.environ.get("DTF_BENCH_BATCH", "2048"))

        def make(c, s_):
            return bench_lib.make_resnet50_strategy(c, bucket_mb=bucket_mb, grad_dtype=gdt, strategy=s_)
        x, y = bench_lib.synthetic_imagenet_batch(batch, dev, size=a.image, seed=rank)
        metric, unit = "images/sec (whole node) ResNet-50 sync DP", "images/sec"
        cfg_model = {"model": "resnet50", "image": a.image, "seq_len": None,
                     "optimizer": "sgd-momentum (fused flat-arena kernel)"}
    else:
        from distributed_tensorflow_amd.models.bert import synthetic_mlm_batch
        strategy = a.strategy or "custom:direct"
        batch = a.batch or 384
        size = "base" if a.model == "bert" else "tiny"
        seq = a.seq_len if size == "base" else min(a.seq_len, 64)

        def make(c, s_):
            return bench_lib.make_bert_strategy(c, size, seq, bucket_mb=bucket_mb, grad_dtype=gdt, strategy=s_)
        vocab = 30522 if size == "base" else 1024
        x, y = synthetic_mlm_batch(batch, seq, vocab, dev, n_pred=min(80, seq // 4), seed=rank)
        metric, unit = "sequences/sec (whole node) BERT-base MLM seq 512", "sequences/sec"
        cfg_model = {"model": "bert-base-mlm" if size == "base" else "bert-tiny-mlm", "image": None, "seq_len": seq,
                     "optimizer": "adam (fused flat-arena kernel)"}

    def sync():
        comm.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    # ---- single-rank reference of the phases: every rank alone, before the process group carries traffic
    ref_phase = {}
    if world > 1 and a.ref_steps > 0 and not a.graph:
        ref = make(Comm(0, 1, dev), "keras")
        for _ in range(2):
            ref.train_step(x, y)
        with PhaseTimer(dev) as pt:
            for _ in range(a.ref_steps):
                ref.train_step(x, y)
            ref_phase = {k: round(v, 3) for k, v in pt.report().items()}
        del ref
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
    st = make(comm, strategy)

    # proof of the process group: every rank contributes a one
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    comm.all_reduce(ones)
    ranks_seen = int(round(float(ones)))

    st.cfg.graph = bool(a.graph and dev.type == "cuda")
    step = st.step_callable()
    feed = None
    if a.input == "host":
        if a.model != "resnet50" or st.cfg.graph:
            print("--input host feeds the eager ResNet-50 step only", file=sys.stderr)
            return 2
        feed = bench_lib.HostInputPipeline(batch, dev, size=a.image, seed=rank)
        train = step

        def step(_x, _y):      # noqa: F811 - every step trains on the batch the pipeline delivers
            return train(*feed.next())
    # untimed priming before the counted warm-up steps: communicator set-up and (with --graph) the eager steps and
    # the capture all happen here, whatever --warmup is
    for _ in range(5 if st.cfg.graph else 2):
        step(x, y)
    for _ in range(a.warmup):
        step(x, y)
    sync()
    t0 = time.perf_counter()
    loss = None
    for _ in range(a.steps):
        loss = step(x, y)
    sync()
    dt = time.perf_counter() - t0
    dt_min = dt
    if world > 1:
        # the slowest rank's time counts; the fastest rank's is reported next to it (the spread between ranks)
        t = torch.tensor([dt, -dt], dtype=torch.float64 if dev.type == "cpu" else torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_min = float(t[0]), -float(t[1])
    loss_v = float(loss)

    # ---- after the timed region: where the time goes, and whether the replicas still agree
    phase_ms, exposed = {}, None
    if a.phase_steps > 0 and not st.cfg.graph:
        with PhaseTimer(dev) as pt:
            sync()
            tp = time.perf_counter()
            for _ in range(a.phase_steps):
                st.train_step(x, y)
            sync()
            step_ms = (time.perf_counter() - tp) * 1e3 / a.phase_steps
            phase_ms = {k: round(v, 3) for k, v in pt.report().items()}
        wait = sum(v for k, v in phase_ms.items()
                   if k in ("grad_sync_wait", "push_wait", "pull", "push_pull", "gossip", "comm_in_hook"))
        exposed = round(wait / step_ms, 4) if step_ms > 0 else None
    # host side of the communication: time inside the bucket-ready callbacks and operations issued per bucket
    red = st.replica.reducer
    hook_ms = round(red.hook_host_s * 1e3 / max(1, st.replica.step_count), 3)
    from distributed_tensorflow_amd.comm import algorithms as _alg
    ops_per_bucket = None
    if red.hook_calls and strategy.startswith("custom"):
        ops_per_bucket = round((_alg.stats["exchanges"] + _alg.stats["kernels"]) / red.hook_calls, 2)
    slowdown = None
    if ref_phase.get("backward") and phase_ms.get("backward"):
        # the hook's share is communication (reported on its own), not a slower backward pass
        slowdown = round((phase_ms["backward"] - phase_ms.get("comm_in_hook", 0.0)) / ref_phase["backward"], 4)
    if feed is not None:
        feed.close()
    st.finalize()
    agree = None
    if st.replicas_agree or strategy.startswith(("graph", "async")):
        try:
            saved = st.replicas_agree
            st.replicas_agree = True       # after finalize() every strategy's replicas hold the same model
            st.check_replicas(a.steps)
            st.replicas_agree = saved
            agree = True
        except Exception as e:  # ReplicaDivergence on every rank
            agree = False
            print(f"rank {rank}: {e}", file=sys.stderr)
    buckets = [b.numel * st.flat.grad.element_size() for b in st.planner.buckets]
    if rank == 0:
        ms = dt / a.steps * 1e3
        value = batch * world * a.steps / dt
        print(json.dumps({
            "metric": metric, "value": round(value, 2), "unit": unit,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "input": "one GPU-resident batch" if feed is None else "host uint8 pool -> loader threads -> pinned -> copy stream",
            "config": {**cfg_model, "global_batch": batch * world, "per_gpu_batch": batch,
                       "parallelism": f"dp{world}", "strategy": strategy, "grad_dtype": a.grad_dtype,
                       "kernels": "own (gemm.hip / gemm256.hip / norm.hip ...): no vendor-library compute route",
                       "hip_graph": st.cfg.graph},
            "ranks_seen": ranks_seen, "comm_backend": comm.backend if world > 1 else "none (single rank)",
            "ms_per_step_fastest_rank": round(dt_min / a.steps * 1e3, 3),
            "buckets": {"count": len(buckets), "cap_mib": bucket_mb, "total_mib": round(sum(buckets) / 2 ** 20, 2),
                        "largest_mib": round(max(buckets) / 2 ** 20, 2)},
            "phase_ms": phase_ms, "exposed_comm_fraction": exposed,
            "phase_ms_single_rank_reference": ref_phase, "backward_slowdown": slowdown,
            "comm_host_ms_per_step": hook_ms, "comm_host_ops_per_bucket": ops_per_bucket,
            "replicas_agree": agree,
            "vram_pretouch": pretouch,
            "final_loss": loss_v,
        }), flush=True)
    comm.shutdown()
    return 0


if __name__ == "__main__":
    sys.exit(main())
