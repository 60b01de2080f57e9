"""Auxiliary subsystems on CPU: checkpoint/resume, transport, barrier, metrics, tracing, loader."""
import json
import os
import sys
import threading
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.comm import tcp  # noqa: E402
from distributed_tensorflow_amd.comm.backend import Comm  # noqa: E402
from distributed_tensorflow_amd.engine.optimizer import Adam  # noqa: E402
from distributed_tensorflow_amd.engine.replica import TrainConfig  # noqa: E402
from distributed_tensorflow_amd.models.mlp import reference_mlp  # noqa: E402
from distributed_tensorflow_amd.obs.metrics import MetricsLogger  # noqa: E402
from distributed_tensorflow_amd.obs.trace import PhaseTimer  # noqa: E402
from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy  # noqa: E402
from distributed_tensorflow_amd.utils.checkpoint import load_checkpoint, save_checkpoint  # noqa: E402
from distributed_tensorflow_amd.utils.sync import BarrierBroken, DynamicBarrier  # noqa: E402


def _strategy(seed=0):
    torch.manual_seed(seed)
    return MirroredStrategy(reference_mlp(dropout=0.0), Comm(0, 1, "cpu"), TrainConfig(), optimizer=Adam())


def test_checkpoint_resume_is_bit_exact(tmp_path):
    x = torch.rand(16, 28, 28, 1); y = torch.randint(0, 10, (16,), dtype=torch.int32)
    a = _strategy()
    for _ in range(3):
        a.train_step(x, y)
    save_checkpoint(str(tmp_path / "ck"), a)
    for _ in range(2):
        a.train_step(x, y)
    b = _strategy(seed=123)                       # different init: everything must come from the checkpoint
    step = load_checkpoint(str(tmp_path / "ck"), b)
    assert step == 3
    for _ in range(2):
        b.train_step(x, y)
    assert torch.equal(a.flat.master, b.flat.master)
    assert torch.equal(a.optimizer.m, b.optimizer.m) and torch.equal(a.optimizer.v, b.optimizer.v)
    # the Keras-ordered export next to it loads into a fresh model
    m = reference_mlp(dropout=0.0)
    m.load_weights(str(tmp_path / "ck" / "weights.npz"))
    assert len(m.get_weights()) == 4


def test_tcp_transport_roundtrip_large_and_timeouts():
    lst = tcp.Listener("127.0.0.1", 0)
    got = {}

    def server():
        c = lst.accept(timeout=10)
        buf = torch.empty(3_000_000, dtype=torch.float32)
        kind, meta, payload = c.recv(into=buf)
        got["msg"] = (kind, meta, payload.clone())
        c.send(7, {"ok": True}, payload * 2)
        kind, meta, payload = c.recv()
        got["empty"] = (kind, meta, payload)
        with pytest.raises(tcp.TransportError):
            c.recv(timeout=0.2)                  # nothing arrives: bounded wait, not a hang
        c.close()

    t = threading.Thread(target=server); t.start()
    c = tcp.connect("127.0.0.1", lst.port, timeout=10)
    data = torch.arange(3_000_000, dtype=torch.float32)
    c.send(3, {"version": 5, "name": "grads"}, data)
    out = torch.empty_like(data)
    kind, meta, payload = c.recv(into=out)
    assert kind == 7 and meta == {"ok": True} and torch.equal(payload, data * 2)
    c.send(4, {})
    t.join(10)
    assert got["msg"][0] == 3 and got["msg"][1] == {"version": 5, "name": "grads"} and torch.equal(got["msg"][2], data)
    assert got["empty"] == (4, {}, None)
    c.close(); lst.close()


def test_connect_retries_until_server_listens_and_gives_up_eventually():
    with pytest.raises(tcp.TransportError):
        tcp.connect("127.0.0.1", 1, timeout=0.5, retry_interval=0.1)     # nobody listens on port 1
    holder = {}

    def late_server():
        time.sleep(0.5)
        holder["l"] = tcp.Listener("127.0.0.1", 6391)
        holder["c"] = holder["l"].accept(timeout=10)

    t = threading.Thread(target=late_server); t.start()
    c = tcp.connect("127.0.0.1", 6391, timeout=10, retry_interval=0.1)   # like the reference's retry loop
    t.join(10)
    c.close(); holder["c"].close(); holder["l"].close()


def test_peer_close_is_reported():
    lst = tcp.Listener("127.0.0.1", 0)
    t = threading.Thread(target=lambda: lst.accept(timeout=10).close()); t.start()
    c = tcp.connect("127.0.0.1", lst.port, timeout=10)
    t.join(10)
    with pytest.raises(tcp.ConnectionClosed):
        c.recv(timeout=5)
    c.close(); lst.close()


def test_dynamic_barrier_action_leave_abort_timeout():
    b = DynamicBarrier(3, timeout=5)
    hits = []
    ts = [threading.Thread(target=lambda: b.wait(lambda: hits.append(1))) for _ in range(3)]
    [t.start() for t in ts]; [t.join(5) for t in ts]
    assert hits == [1]                                   # the action ran exactly once
    ts = [threading.Thread(target=lambda: b.wait(lambda: hits.append(2))) for _ in range(2)]
    [t.start() for t in ts]
    time.sleep(0.1)
    b.leave()                                            # third party departs -> round completes
    [t.join(5) for t in ts]
    assert hits == [1, 2] and b.parties == 2
    errs = []

    def waiter():
        try:
            b.wait()
        except BarrierBroken as e:
            errs.append(e)
    t = threading.Thread(target=waiter); t.start(); time.sleep(0.1); b.abort(); t.join(5)
    assert len(errs) == 1
    b2 = DynamicBarrier(2, timeout=0.2)
    with pytest.raises(BarrierBroken):
        b2.wait()


def test_metrics_jsonl_and_phase_timer(tmp_path):
    path = str(tmp_path / "m_%r.jsonl")
    m = MetricsLogger(path, rank=3)
    m.log(step=0, loss=1.5)
    m.summary(elapsed=2.0)
    m.close()
    rows = [json.loads(l) for l in open(str(tmp_path / "m_3.jsonl"))]
    assert rows[0]["step"] == 0 and rows[0]["rank"] == 3 and rows[1]["event"] == "summary"
    st = _strategy()
    x = torch.rand(8, 28, 28, 1); y = torch.randint(0, 10, (8,), dtype=torch.int32)
    with PhaseTimer("cpu") as pt:
        st.train_step(x, y)
        rep = pt.report()
    assert {"forward", "backward", "optimizer"} <= set(rep)


def test_native_loader_gathers_and_normalises():
    from distributed_tensorflow_amd.data.loader import NativeLoader
    x = np.arange(50 * 12, dtype=np.uint8).reshape(50, 3, 4)
    y = np.arange(50, dtype=np.int32)
    ld = NativeLoader(x, y, max_batch=8, to_float=True, slots=3, threads=2, pinned=False)
    idx = [np.array([5, 0, 49, 7]), np.array([1, 2, 3, 4, 5, 6, 7, 8]), np.array([10])]
    slots = [ld.submit(i) for i in idx]
    for s, i in zip(slots, idx):
        bx, by = ld.take(s)
        assert bx.shape == (len(i), 3, 4) and bx.dtype == torch.float32
        assert torch.allclose(bx, torch.from_numpy(x[i].astype(np.float32) / 255.0))
        assert by.tolist() == y[i].tolist()
        ld.release(s)
    ld.close()


# ------------------------------------------------------------------ peer-pointer schedules
@pytest.mark.parametrize("algorithm", ["direct", "oneshot", "ring", "tree"])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 7, 8])
def test_peer_pointer_schedule_emulation(algorithm, world):
    """Every rank ends with scale * sum, and no phase reads what another rank writes."""
    from distributed_tensorflow_amd.comm import ipc

    torch.manual_seed(world)
    for n in (5, 64, 1003):
        bufs = [torch.randn(n, dtype=torch.float32) for _ in range(world)]
        want = sum(b.double() for b in bufs) * 0.5
        ipc.emulate(algorithm, bufs, scale=0.5)
        for r, b in enumerate(bufs):
            assert torch.allclose(b.double(), want, atol=1e-5), (algorithm, world, n, r)


def test_peer_pointer_emulator_detects_races():
    from distributed_tensorflow_amd.comm import ipc

    real_plan = ipc.plan

    def racy(algorithm, rank, world, n, esz, scale=1.0):
        # a "direct" schedule without the barrier between reduce-scatter and all-gather
        ph = real_plan("direct", rank, world, n, esz, scale)
        return [ph[0] + ph[1]]

    ipc.plan = racy
    try:
        with pytest.raises(AssertionError):
            ipc.emulate("direct", [torch.randn(64) for _ in range(4)])
    finally:
        ipc.plan = real_plan
