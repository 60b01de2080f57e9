"""Single-GPU training throughput of the non-headline model configurations.

usage: python tools/bench_models.py [bert|resnet20|mlp|widedeep ...] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.comm.backend import Comm  # noqa: E402
from distributed_tensorflow_amd.engine.optimizer import SGD, Adam  # noqa: E402
from distributed_tensorflow_amd.engine.replica import TrainConfig  # noqa: E402
from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy  # noqa: E402

DEV = torch.device("cuda:0")


def run(name, st, x, y, steps, warmup, unit_count, unit, graph=False):
    st.cfg.graph = graph
    step = st.step_callable()
    for _ in range(warmup + (5 if graph else 0)):
        step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {"model": name, "hip_graph": graph, "ms_per_step": round(dt * 1e3, 3), unit: round(unit_count / dt, 1),
           "loss": float(loss)}
    print(json.dumps(res), flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("models", nargs="*", default=["bert", "resnet20", "mlp"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--bert-batch", type=int, default=64)
    a = ap.parse_args()
    comm = Comm(0, 1, DEV)
    out = []
    if "bert" in a.models:
        from distributed_tensorflow_amd.models.bert import bert_base, synthetic_mlm_batch
        torch.manual_seed(0)
        cfg = TrainConfig(loss="mlm", param_dtype=torch.bfloat16, grad_dtype=torch.bfloat16)
        st = MirroredStrategy(bert_base(512), comm, cfg, optimizer=Adam(learning_rate=1e-4))
        x, y = synthetic_mlm_batch(a.bert_batch, 512, 30522, DEV, n_pred=80)
        out.append(run("bert-base-mlm-seq512", st, x, y, a.steps, a.warmup, a.bert_batch, "sequences_per_s"))
        try:
            out.append(run("bert-base-mlm-seq512", st, x, y, a.steps, a.warmup, a.bert_batch, "sequences_per_s", True))
        except Exception as e:  # a step that cannot be captured is reported, not fatal
            out.append({"model": "bert-base-mlm-seq512", "hip_graph": True, "error": str(e)[:300]})
            print(out[-1], flush=True)
        del st
    if "resnet20" in a.models:
        from distributed_tensorflow_amd.models.resnet import resnet20
        torch.manual_seed(0)
        cfg = TrainConfig(param_dtype=torch.bfloat16, grad_dtype=torch.bfloat16)
        st = MirroredStrategy(resnet20(), comm, cfg, optimizer=SGD(0.1, momentum=0.9, formula="torch"))
        for bs in (1024, 128):
            x = torch.randn(bs, 32, 32, 3, device=DEV).to(torch.bfloat16)
            y = torch.randint(0, 10, (bs,), device=DEV)
            for graph in (False, True):
                out.append(run(f"resnet20-cifar10-b{bs}", st, x, y, a.steps * 3, a.warmup, bs, "images_per_s", graph))
                st._graphed = None
        del st
    if "mlp" in a.models:
        from distributed_tensorflow_amd.models.mlp import reference_mlp
        torch.manual_seed(0)
        for dt, bs in ((None, 100), (torch.bfloat16, 100), (torch.bfloat16, 8192)):
            cfg = TrainConfig(param_dtype=dt)
            x = torch.rand(bs, 28, 28, 1, device=DEV)
            y = torch.randint(0, 10, (bs,), device=DEV, dtype=torch.int32)
            for graph in (False, True):
                st = MirroredStrategy(reference_mlp(), comm, cfg, optimizer=Adam())
                out.append(run(f"reference-mlp-b{bs}-{'bf16' if dt else 'fp32'}", st, x, y, a.steps * 20, a.warmup, bs,
                               "images_per_s", graph))
        # parameter-server step latency (push = reduce-scatter, fused apply on the shard, pull = all-gather)
        from distributed_tensorflow_amd.parallel.ps import SyncPSStrategy
        x = torch.rand(64, 28, 28, 1, device=DEV)
        y = torch.randint(0, 10, (64,), device=DEV, dtype=torch.int32)
        for graph in (False, True):
            st = SyncPSStrategy(reference_mlp(), comm, TrainConfig(loss_reduction="sum"), optimizer=Adam())
            out.append(run("ps-sync-step-latency-mlp-b64-fp32", st, x, y, a.steps * 20, a.warmup, 64, "images_per_s",
                           graph))
    if "widedeep" in a.models:
        from distributed_tensorflow_amd.models.wide_deep import bench_step
        out.append(bench_step(DEV, a.steps, a.warmup))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_models.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
