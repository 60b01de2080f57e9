"""BASELINE config 5 with all of its shards on ONE MI355X: N processes share the GPU, each owns 1/N of the rows of
the two embedding tables (deep 128 wide, wide 4 wide) with their Adagrad state, the dense part is replicated and
trained through the synchronous parameter server. gloo carries the collectives (RCCL refuses several ranks on one
device; every transfer is staged through host memory, so the step time says nothing about xGMI) -- what the run
shows is the LAYOUT at full size: --state-gb of sharded server state resident in HBM, the id / row / gradient
exchange between owners at the real batch size, one routing plan per step for both tables, replicated dense
weights that stay bit-identical.

usage: python tools/bench_widedeep_ranks.py [--ranks 8] [--state-gb 200] [--batch 8192] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q, a):
    import traceback
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from distributed_tensorflow_amd.comm.backend import Comm
        from distributed_tensorflow_amd.engine.optimizer import Adam
        from distributed_tensorflow_amd.engine.replica import TrainConfig
        from distributed_tensorflow_amd.models import wide_deep as wd
        from distributed_tensorflow_amd.parallel.ps import SyncPSStrategy
        dev = torch.device("cuda", 0) if a.device == "gpu" else torch.device("cpu")
        if dev.type == "cuda":
            torch.cuda.set_device(dev)
        else:
            torch.set_num_threads(1)
        comm = Comm.init(rank, world, dev, "127.0.0.1", 0, timeout_s=600, backend="gloo", port_cell=port)
        emb_dim, wide_dim = 128, 4
        # state per row: table + Adagrad accumulator, both tables, fp32
        rows = int(a.state_gb * (1 << 30) / (2 * 4 * (emb_dim + wide_dim)))
        rows -= rows % world
        torch.manual_seed(0)
        model = wd.WideDeep(comm, rows, emb_dim=emb_dim, wide_dim=wide_dim)
        cfg = TrainConfig(loss=wd.bce_loss, loss_reduction="mean",
                          param_dtype=torch.bfloat16 if dev.type == "cuda" else torch.float32)
        st = SyncPSStrategy(model, comm, cfg, optimizer=Adam(learning_rate=1e-3))
        x, y = wd.synthetic_batch(a.batch, rows, model.n_sparse, model.n_dense, dev, seed=rank)
        losses = []
        for _ in range(a.warmup):
            losses.append(float(st.train_step(x, y)))
        comm.barrier()
        sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            losses.append(float(st.train_step(x, y)))
        sync()
        comm.barrier()
        dt = (time.perf_counter() - t0) / a.steps
        st.finalize()
        dense = torch.cat([p.detach().float().reshape(-1) for p in model.mlp.parameters()])
        q.put((rank, {"ms_per_step": dt * 1e3, "losses": losses,
                      "shard_state_gb": sum(m.nbytes() for m in model.sparse_modules()) / (1 << 30),
                      "rows": rows, "rows_pulled": model.deep_emb.rows_pulled,
                      "allocated_gb": torch.cuda.max_memory_allocated(dev) / (1 << 30) if dev.type == "cuda" else 0.0,
                      "dense_checksum": float(dense.double().sum()), "dense_absmax": float(dense.abs().max())}))
        comm.barrier()
        comm.shutdown()
    except BaseException:
        q.put((rank, "ERR " + traceback.format_exc()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--state-gb", type=float, default=200.0, help="sharded server state over all ranks (tables + Adagrad)")
    ap.add_argument("--batch", type=int, default=8192, help="examples per rank and step")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--device", default="gpu", choices=["gpu", "cpu"], help="cpu: plumbing check of this tool")
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), ctx.Value("i", 0)
    procs = [ctx.Process(target=worker, args=(r, a.ranks, port, q, a)) for r in range(a.ranks)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(a.ranks):
        r, v = q.get(timeout=900)
        if isinstance(v, str):
            print(f"rank {r}: {v}", file=sys.stderr)
            for p in procs:
                p.terminate()
            return 1
        out[r] = v
        print(f"rank {r} done", flush=True)
    for p in procs:
        p.join(120)
    ms = max(v["ms_per_step"] for v in out.values())
    res = {"config": "wide_deep, row-sharded tables, sync PS, all shards on one GPU (gloo, host-staged transfers)",
           "ranks": a.ranks, "batch_per_rank": a.batch, "rows": out[0]["rows"],
           "sharded_state_gb_total": round(sum(v["shard_state_gb"] for v in out.values()), 2),
           "sharded_state_gb_per_rank": round(out[0]["shard_state_gb"], 2),
           "gpu_allocated_gb_sum": round(sum(v["allocated_gb"] for v in out.values()), 2),
           "ms_per_step_slowest_rank": round(ms, 2), "examples_per_s": round(a.batch * a.ranks / ms * 1e3, 1),
           "rows_pulled_per_rank": out[0]["rows_pulled"],
           "loss_first_last": [round(out[0]["losses"][0], 5), round(out[0]["losses"][-1], 5)],
           "dense_replicas_identical": len({v["dense_checksum"] for v in out.values()}) == 1}
    print(json.dumps(res), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"widedeep_ranks{a.ranks}.json"), "w") as f:
        json.dump(res, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
