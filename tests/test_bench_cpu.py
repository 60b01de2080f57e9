"""bench.py contract: one JSON line from rank 0, launched the way the driver launches it."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, OMP_NUM_THREADS="2", PYTHONPATH=ROOT)


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(line, n):
    r = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["n_gpus"] == n and r["steps"] == 2 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["dtype"] == "bf16" and r["data"] == "synthetic"
    assert r["config"]["model"] == "resnet50" and r["config"]["global_batch"] == 2 * n
    assert r["config"]["parallelism"] == f"dp{n}"
    assert r["value"] > 0 and abs(r["value"] - 2 * n / (r["ms_per_step"] / 1e3)) / r["value"] < 0.01
    for k in ("ranks_seen", "comm_backend", "buckets", "phase_ms", "exposed_comm_fraction", "replicas_agree"):
        assert k in r, k
    assert r["ranks_seen"] == n
    assert r["buckets"]["count"] >= 1 and r["buckets"]["cap_mib"] == 48
    return r


@pytest.mark.slow
def test_bench_single_process_cpu_plumbing():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--batch", "2", "--image", "32",
                        "--steps", "2", "--warmup", "1"], env=ENV, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    _check(lines[0], 1)


def test_bench_host_input_pipeline_cpu_plumbing():
    """--input host: every step trains on a batch that the native loader gathered from the host pool."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--batch", "2", "--image", "32",
                        "--steps", "2", "--warmup", "1", "--input", "host", "--phase-steps", "0"], env=ENV,
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["input"].startswith("host") and r["value"] > 0 and r["final_loss"] == r["final_loss"]


def test_host_input_pipeline_delivers_the_pool_rows():
    import numpy as np
    import torch

    sys.path.insert(0, ROOT)
    from distributed_tensorflow_amd.bench_lib import HostInputPipeline
    feed = HostInputPipeline(4, "cpu", size=8, classes=10, pool=16, seed=3)
    seen = []
    for _ in range(5):
        x, y = feed.next()
        assert x.shape == (4, 8, 8, 3) and x.dtype == torch.bfloat16 and y.shape == (4,)
        # every delivered image is a row of the pool, normalised, with that row's label
        raw = (x.float() * 64.0 + 127.5).round().clamp(0, 255).to(torch.uint8).numpy()
        for i in range(4):
            match = [j for j in range(16) if int(feed._y[j]) == int(y[i])
                     and np.abs(raw[i].astype(int) - feed._x[j].astype(int)).max() <= 1]
            assert match, "delivered image is not a pool row"
        seen.append(y.clone())
    assert any(not torch.equal(seen[0], s) for s in seen[1:])       # batches differ from step to step
    feed.close()


@pytest.mark.slow
def test_bench_two_ranks_under_torchrun_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device", "cpu", "--batch", "2",
           "--image", "32", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # only rank 0 prints
    _check(lines[0], 2)


_TWO = [(s, 2) for s in ["keras", "custom:ring", "custom:direct", "sync", "async", "graph:1"]]
_NODE = [(s, 8) for s in ["keras", "custom:direct", "custom:multiring", "sync", "async", "graph:3"]]


@pytest.mark.parametrize("strategy,world", _TWO + _NODE)
def test_bench_every_strategy_two_ranks_and_eight(strategy, world):
    """The benchmark drives every strategy; the JSON proves that all ranks took part, shows the phases and reports
    that the replicas agree after the run. World 8 is the launch the driver makes on a whole node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--device", "cpu",
           "--model", "bert-tiny", "--batch", "2", "--steps", "2", "--warmup", "1", "--strategy", strategy,
           "--comm-timeout-s", "120"]
    p = subprocess.run(cmd, env=dict(ENV, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["ranks_seen"] == world and r["n_gpus"] == world and r["comm_backend"] == "gloo"
    assert r["config"]["strategy"] == strategy and r["config"]["model"] == "bert-tiny-mlm"
    assert r["unit"] == "sequences/sec" and r["value"] > 0
    assert r["replicas_agree"] is True
    assert "forward" in r["phase_ms"] and "backward" in r["phase_ms"]
    if strategy in ("keras", "sync"):
        assert r["exposed_comm_fraction"] is not None
    # contention evidence: the same phases on one rank alone, and where the collectives are started
    assert "backward" in r["phase_ms_single_rank_reference"] and r["backward_slowdown"] > 0
    if strategy != "async" and not strategy.startswith("graph"):
        assert "comm_in_hook" in r["phase_ms"] and r["comm_host_ms_per_step"] > 0
    if strategy == "custom:direct":
        assert r["comm_host_ops_per_bucket"] == 3.0         # two grouped exchanges + one reduction, at any world size
    if strategy == "custom:ring":
        assert r["comm_host_ops_per_bucket"] == 3.0 * (world - 1)


def test_collective_harness_two_ranks_gloo():
    """tools/bench_collectives.py over gloo: every back end that applies on the CPU (process-group all-reduce, the
    five point-to-point schedules, reduce-scatter + all-gather) reproduces the exact sum and is timed."""
    out = os.path.join(ROOT, "gpurun_out", "collectives_w2.json")
    if os.path.exists(out):
        os.remove(out)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_collectives.py"), "--gpus", "2", "--device", "cpu",
           "--sizes-mib", "0.25", "--dtype", "fp32", "--iters", "2"]
    p = subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    with open(out) as f:
        d = json.load(f)
    assert d["world"] == 2 and d["backend"] == "gloo"
    names = {r["backend"] for r in d["results"]}
    assert {"rccl", "custom:direct", "custom:ring", "custom:multiring", "custom:tree", "custom:oneshot", "rs+ag"} <= names
    assert all(r["correct"] and r["us"] > 0 for r in d["results"])


def test_wide_and_deep_rank_tool_runs_on_cpu():
    """tools/bench_widedeep_ranks.py (config 5 with all shards as processes of one device): plumbing mode, 2 ranks."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_widedeep_ranks.py"), "--device", "cpu", "--ranks", "2",
           "--state-gb", "0.01", "--batch", "64", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["ranks"] == 2 and r["dense_replicas_identical"] and r["rows_pulled_per_rank"] == 3 * 64 * 26
    assert r["loss_first_last"][1] < r["loss_first_last"][0]
