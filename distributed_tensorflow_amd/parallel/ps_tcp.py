"""Parameter server and worker over the framed-TCP transport.

This is the centralized topology in the shape the reference gives it: one
server process listening on ``server_address:6000`` with one service thread
per worker connection, and ``n_nodes`` worker processes (reference
centralized/server.py:139-187, centralized/client.py:45-101). It is the
multi-machine path (``-tt server`` / ``-tt worker -ti k -sa ADDR``) and the
CPU plumbing mode; on a single GPU node the collective strategies of
``parallel/ps.py`` replace it.

Protocol, per worker connection (kinds below):

    worker: START {index}                      server: barrier; WEIGHTS(flat fp32 master)
    worker: TRAIN {version, loss}(flat grads)  server: apply; [sync: barrier]; WEIGHTS
    worker: END                                server: barrier; thread exits

Differences from the reference, all deliberate:
  * tensors travel as raw bytes of the flat arenas, not pickled lists;
  * ``sync`` + ``apply='reduced'`` sums the N gradients and takes ONE optimizer step;
    ``apply='sequential'`` is the reference behaviour, one apply per worker under the lock, each
    with its own optimizer state unless one optimizer is shared (server.py:52-55,91-93);
  * the weights sent after a synchronous round are a single snapshot taken when the barrier
    trips, so no worker can observe a half-applied next round (server.py:95-96 reads outside
    the lock);
  * barriers have a timeout and are aborted when a connection drops, so a dead worker ends
    the job with an error instead of hanging it (server.py:132-134 keeps looping);
  * the server replies with trainable parameters only; buffers such as batch-norm statistics
    stay local to each worker (server.py:96,101 overwrites them);
  * the server, which owns the model, also owns its persistence: it imports / exports the Keras-ordered
    weight list and writes and resumes checkpoints (weights, optimizer state, per-worker step counts); a
    worker adds its own random-number state and, on resume, skips the batches its completed steps consumed.
"""
from __future__ import annotations

import inspect
import os
import threading
import time

import torch

from ..api.model import as_model
from ..comm import tcp
from ..engine.optimizer import FlatOptimizer
from ..engine.replica import Replica, TrainConfig
from ..obs.metrics import MetricsLogger, SupervisorSink
from ..utils.affinity import pin_to_core
from ..utils.sync import BarrierBroken, DynamicBarrier

START, WEIGHTS, TRAIN, END, ERROR = 1, 2, 3, 4, 5
DEFAULT_PORT = 6000


def call_dataset_fn(dataset_fn, batch_size, n_nodes=None, **kw):
    """Calls the user's ``dataset_fn`` with the reference's keyword set, adding ``n_nodes`` when
    the function accepts it (the reference reads it from a module global that only exists in
    forked children, reference initializer.py:44,119)."""
    try:
        params = inspect.signature(dataset_fn).parameters
        accepts = "n_nodes" in params or any(p.kind == p.VAR_KEYWORD for p in params.values())
        kw = {k: v for k, v in kw.items() if k in params or any(p.kind == p.VAR_KEYWORD for p in params.values())}
    except (TypeError, ValueError):
        accepts = False
    if n_nodes is not None:
        os.environ["DTF_N_NODES"] = str(n_nodes)
        if accepts:
            kw["n_nodes"] = n_nodes
    return dataset_fn(batch_size, **kw)


class _ServerState:
    def __init__(self, replica: Replica, n_nodes: int, sync_mode: str, apply: str, optimizer, barrier_timeout: float):
        self.replica = replica
        self.flat = replica.flat
        self.n = n_nodes
        self.sync_mode = sync_mode
        self.apply = apply
        self.lock = threading.Lock()
        self.barrier = DynamicBarrier(n_nodes, timeout=barrier_timeout)
        shared = optimizer is not None or apply == "reduced" or sync_mode == "async"
        self.shared_opt = FlatOptimizer(self.flat, optimizer) if shared else None
        self.optimizer_spec = optimizer
        self.sum_grad = torch.zeros_like(self.flat.grad) if (sync_mode == "sync" and apply == "reduced") else None
        self.snapshot = self.flat.master.detach().cpu().clone()
        self.version = 0
        self.start_time = 0.0
        self.elapsed = None
        self.ended = 0
        self.staleness: list[int] = []
        self.staleness_lr = os.environ.get("DTF_ASYNC_STALENESS_LR", "1") != "0"
        self.step_latency: list[float] = []
        self.errors: list[BaseException] = []
        self.steps_by_worker: dict[int, int] = {}      # worker index -> steps the server has applied for it
        self.worker_opts: dict[int, FlatOptimizer] = {}   # per-worker optimizer states (apply='sequential')
        self.checkpoint_dir: str | None = None
        self.checkpoint_every = 0
        self.resumed_from: dict[int, int] = {}
        self._pending_opt_state: dict[int, dict] = {}
        self.checkpoints_written = 0

    # barrier actions: run by exactly one thread while the others are still blocked
    def on_all_started(self):
        if self.start_time == 0.0:
            self.start_time = time.time()

    def on_round_complete(self):
        if self.sum_grad is not None:
            self.shared_opt.step(grad=self.sum_grad)
            self.sum_grad.zero_()
            self.version += 1
        self.snapshot = self.flat.master.detach().cpu().clone()
        self.rounds = getattr(self, "rounds", 0) + 1
        if self.checkpoint_dir and self.checkpoint_every and self.rounds % self.checkpoint_every == 0:
            self.save_checkpoint()

    # ------------------------------------------------------------ persistence (the server owns the model)
    def save_checkpoint(self) -> bool:
        """Weights, optimizer state(s), version and per-worker step counts, written atomically. The caller is the only
        thread touching the model (the barrier action, a holder of ``lock``, or the main thread after the join)."""
        from ..utils.checkpoint import _atomic_save
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        flat = self.flat
        slots = {s.name: {"offset": s.offset, "numel": s.numel, "shape": list(s.shape)} for s in flat.slots}
        opts = {"shared": self.shared_opt.state_dict() if self.shared_opt is not None else None,
                "per_worker": {int(k): o.state_dict() for k, o in self.worker_opts.items()}}
        _atomic_save({"master": flat.master.detach().cpu().clone(), "slots": slots, "optimizers": opts,
                      "version": int(self.version), "rounds": int(getattr(self, "rounds", 0)),
                      "steps_by_worker": {int(k): int(v) for k, v in self.steps_by_worker.items()},
                      "sync_mode": self.sync_mode, "apply": self.apply}, os.path.join(self.checkpoint_dir, "ps_server.pt"))
        self.checkpoints_written += 1
        return True

    def load_checkpoint(self) -> bool:
        path = os.path.join(self.checkpoint_dir or "", "ps_server.pt")
        if not self.checkpoint_dir or not os.path.exists(path):
            return False
        ck = torch.load(path, map_location="cpu", weights_only=True)
        flat = self.flat
        for s in flat.slots:
            src = ck["slots"][s.name]
            if src["numel"] != s.numel or list(src["shape"]) != list(s.shape):
                raise ValueError(f"checkpoint tensor {s.name} has shape {src['shape']}, model expects {list(s.shape)}")
            flat.master[s.offset:s.offset + s.numel].copy_(ck["master"][src["offset"]:src["offset"] + s.numel])
        flat.refresh_shadow()
        if self.shared_opt is not None and ck["optimizers"]["shared"] is not None:
            self.shared_opt.load_state_dict(ck["optimizers"]["shared"])
        self._pending_opt_state = {int(k): v for k, v in ck["optimizers"]["per_worker"].items()}
        self.version = int(ck["version"])
        self.rounds = int(ck.get("rounds", 0))
        self.steps_by_worker = {int(k): int(v) for k, v in ck["steps_by_worker"].items()}
        self.resumed_from = dict(self.steps_by_worker)
        self.snapshot = flat.master.detach().cpu().clone()
        return True

    def optimizer_for(self, index: int) -> FlatOptimizer:
        """The optimizer state that applies worker ``index``'s gradients: the shared one, or that worker's own
        (the reference creates one Adam per connection thread, server.py:52-55)."""
        if self.shared_opt is not None:
            return self.shared_opt
        with self.lock:
            o = self.worker_opts.get(index)
            if o is None:
                o = FlatOptimizer(self.flat, self.optimizer_spec)
                sd = self._pending_opt_state.pop(index, None)
                if sd is not None:
                    o.load_state_dict(sd)
                self.worker_opts[index] = o
        return o

    def count_step(self, index: int) -> None:
        """Under ``lock``: one more gradient of worker ``index`` applied; asynchronous mode checkpoints by count."""
        self.steps_by_worker[index] = self.steps_by_worker.get(index, 0) + 1
        if (self.sync_mode == "async" and self.checkpoint_dir and self.checkpoint_every
                and sum(self.steps_by_worker.values()) % (self.checkpoint_every * self.n) == 0):
            self.save_checkpoint()

    def worker_ended(self):
        with self.lock:
            self.ended += 1
            if self.ended == self.n and self.elapsed is None:
                self.elapsed = time.time() - self.start_time


class _ServiceThread(threading.Thread):
    """One per worker connection: the reference's ``Server(threading.Thread)`` (server.py:40-136)."""

    def __init__(self, conn: tcp.Connection, st: _ServerState, wid: int):
        super().__init__(daemon=True, name=f"ps-conn-{wid}")
        self.conn, self.st, self.wid = conn, st, wid
        self.index = wid             # replaced by the worker's own index when its START message arrives
        self.opt = None
        self.grad = torch.empty_like(st.flat.grad, device="cpu")

    def run(self):
        st = self.st
        left = False
        try:
            while True:
                kind, meta, payload = self.conn.recv(into=self.grad)
                if kind == START:
                    print("start", flush=True)
                    self.index = int(meta.get("index", self.wid))
                    self.opt = st.optimizer_for(self.index)
                    st.barrier.wait(st.on_all_started)   # all workers said start
                    st.barrier.wait()                    # all ready to receive
                    self.conn.send(WEIGHTS, {"version": st.version, "n": st.flat.total,
                                             "steps_done": int(st.resumed_from.get(self.index, 0))}, st.snapshot)
                elif kind == TRAIN:
                    t0 = time.time()
                    if self.opt is None:
                        self.opt = st.optimizer_for(self.index)
                    written = st.checkpoints_written
                    g = self.grad.to(st.flat.device) if st.flat.device.type != "cpu" else self.grad
                    if st.sync_mode == "sync":
                        with st.lock:
                            if st.sum_grad is not None:
                                st.sum_grad.add_(g)
                            else:
                                self.opt.step(grad=g)
                                st.version += 1
                            st.count_step(self.index)
                        st.barrier.wait(st.on_round_complete)  # the completing thread takes ONE snapshot
                        reply = st.snapshot
                    else:
                        with st.lock:
                            stale = max(0, st.version - int(meta.get("version", st.version)))
                            st.staleness.append(stale)
                            if st.staleness_lr:       # lr / (1 + staleness): see parallel/ps.py AsyncPSStrategy
                                self.opt.set_lr(float(self.opt.spec.learning_rate) / (1.0 + stale))
                            self.opt.step(grad=g)
                            st.version += 1
                            st.count_step(self.index)
                            reply = st.flat.master.detach().cpu().clone()
                    # "checkpoint": the server has just written one that includes this step -- the worker adds its
                    # own random-number state so that a resumed run draws the same dropout masks
                    self.conn.send(WEIGHTS, {"version": st.version,
                                             "checkpoint": bool(st.checkpoints_written != written
                                                                and st.sync_mode == "sync")}, reply)
                    st.step_latency.append(time.time() - t0)
                elif kind == END:
                    print("end", flush=True)
                    st.worker_ended()
                    st.barrier.leave()
                    left = True
                    break
                else:
                    print("Received unknown message", flush=True)
        except tcp.ConnectionClosed as e:
            print("Connection closed", flush=True)
            st.errors.append(e)
            if st.sync_mode == "sync":
                st.barrier.abort()       # a synchronous round can never complete without this worker
            elif not left:
                st.worker_ended()        # asynchronous mode tolerates a departed worker
        except BarrierBroken as e:
            st.errors.append(e)
        except BaseException as e:
            st.errors.append(e)
            st.barrier.abort()
        finally:
            self.conn.close()


def start(sync_mode, model_fn, dataset_fn, n_nodes, optimizer=None, supervisor_address=None, cpu_affinity=False,
          server_address="127.0.0.1", port: int = DEFAULT_PORT, apply: str = "reduced", device="cpu",
          cfg: TrainConfig | None = None, barrier_timeout: float = 600.0, accept_timeout: float = 600.0,
          result_queue=None, metrics_path: str | None = None, evaluate: bool = True, ready_event=None,
          port_cell=None, load_weights: str | None = None, save_weights: str | None = None,
          checkpoint_dir: str | None = None, checkpoint_every: int = 0, resume: bool = False) -> dict:
    """Runs the parameter server until every worker has sent ``end``; then evaluates on the test
    split and prints the reference's two result lines. Signature follows reference
    centralized/server.py:139-140. ``port_cell`` (a shared ``multiprocessing.Value('i')``): the port the
    listener is bound to is published there -- with ``port=0`` the operating system picks it, so two jobs
    on one host never collide.

    ``load_weights`` / ``save_weights``: Keras-ordered weight list the model starts from / is exported to after
    training (the reference's wire format, server.py:96,150). ``checkpoint_dir`` + ``checkpoint_every=N``: a
    checkpoint every N rounds (sync) or N applies per worker (async) and at the end; ``resume`` continues from it:
    every worker is told how many of its steps the checkpoint contains and skips that many batches."""
    if sync_mode not in ("sync", "async"):
        raise ValueError("Invalid centralized strategy")
    if cpu_affinity:
        pin_to_core(0)
    cfg = cfg or TrainConfig(loss_reduction="sum")
    torch.manual_seed(cfg.seed)
    replica = Replica(as_model(model_fn()), device, cfg)
    replica.reducer.detach()
    if load_weights:
        replica.model.load_weights(load_weights)
        replica.flat.refresh_shadow()
    st = _ServerState(replica, n_nodes, sync_mode, apply, optimizer, barrier_timeout)
    st.checkpoint_dir, st.checkpoint_every = checkpoint_dir, int(checkpoint_every or 0)
    resumed = st.load_checkpoint() if resume else False
    sup = SupervisorSink(supervisor_address)
    sup.send("start")
    listener = tcp.Listener(server_address, port, backlog=50)
    if port_cell is not None:
        port_cell.value = int(listener.port)
    if ready_event is not None:
        ready_event.set()
    print("accepting", flush=True)
    threads = []
    try:
        for wid in range(n_nodes):
            conn = listener.accept(timeout=accept_timeout, conn_timeout=barrier_timeout)
            t = _ServiceThread(conn, st, wid)
            t.start()
            threads.append(t)
        for t in threads:
            t.join()
    finally:
        listener.close()
    fatal = [e for e in st.errors if not (sync_mode == "async" and isinstance(e, tcp.ConnectionClosed))]
    if fatal:
        res = {"ok": False, "error": repr(fatal[0])}
        if result_queue is not None:
            result_queue.put(("server", dict(res)))
        raise RuntimeError(f"parameter server stopped: {fatal[0]!r}")
    print("elapsed time = " + str(st.elapsed), flush=True)
    sup.send(["done", st.elapsed])
    if st.elapsed is None:
        st.elapsed = time.time() - st.start_time
    if checkpoint_dir:
        st.save_checkpoint()
    if save_weights:
        replica.flat.refresh_shadow()
        replica.model.save_weights(save_weights)
    res = {"ok": True, "elapsed": st.elapsed, "versions": st.version, "departed_workers": len(st.errors),
           "resumed": bool(resumed), "checkpoints_written": st.checkpoints_written,
           "mean_staleness": sum(st.staleness) / max(1, len(st.staleness)), "max_staleness": max(st.staleness, default=0),
           "ps_step_latency_ms": 1e3 * sum(st.step_latency) / max(1, len(st.step_latency))}
    if evaluate:
        replica.flat.refresh_shadow()
        test_ds = call_dataset_fn(dataset_fn, 100, n_nodes=n_nodes, type="test", shard=False)
        loss, acc, _ = replica.evaluate(test_ds)
        print("Test set accuracy: {:.3%}".format(acc), flush=True)
        sup.send(["results", acc])
        res.update(test_accuracy=acc, test_loss=loss)
    sup.close()
    if metrics_path:
        m = MetricsLogger(metrics_path, rank=-1)
        m.summary(role="server", **res)
        m.close()
    if result_queue is not None:
        result_queue.put(("server", dict(res)))
    res["weights"] = replica.flat.master.detach().cpu().clone()
    return res


def train(batch_size, model_fn=None, dataset_fn=None, num_epochs=1, index=0, n_nodes=0, cpu_affinity=False,
          server_address="127.0.0.1", port: int = DEFAULT_PORT, device="cpu", cfg: TrainConfig | None = None,
          connect_timeout: float = 120.0, io_timeout: float = 600.0, max_steps: int | None = None,
          result_queue=None, fault_hook=None, metrics_path: str | None = None, port_cell=None,
          checkpoint_dir: str | None = None, resume: bool = False, watchdog_s: float = 0.0) -> dict:
    """A parameter-server worker. Signature follows reference centralized/client.py:45-46. ``port_cell``:
    wait for the server to publish its port there (see :func:`start`). ``checkpoint_dir``: where this worker keeps
    its random-number state next to the server's checkpoint; ``resume``: restore it and skip the batches of the
    steps the server's checkpoint already contains. ``watchdog_s``: end the process (exit code 19) when a step
    makes no progress for that long."""
    if cpu_affinity:
        pin_to_core(index + 1)  # core 0 is the parameter server's
    batch_size = int(batch_size)
    if port_cell is not None:
        t_end = time.time() + connect_timeout
        while port_cell.value == 0:
            if time.time() > t_end:
                raise TimeoutError("the parameter server did not publish its port")
            time.sleep(0.01)
        port = int(port_cell.value)
    conn = tcp.connect(server_address, port, timeout=connect_timeout, retry_interval=0.2, conn_timeout=io_timeout)
    cfg = cfg or TrainConfig(loss_reduction="sum")
    torch.manual_seed(cfg.seed + 1 + index)
    dataset = call_dataset_fn(dataset_fn, batch_size, n_nodes=n_nodes or None, index=index)
    replica = Replica(as_model(model_fn()), device, cfg)
    flat = replica.flat
    metrics = MetricsLogger(metrics_path, rank=index)
    host = torch.empty(flat.total, dtype=torch.float32)
    conn.send(START, {"index": index})
    kind, meta, _ = conn.recv(into=host)
    if kind != WEIGHTS:
        raise RuntimeError(f"expected initial weights, got message kind {kind}")
    flat.load_flat(host.to(flat.device))
    version = int(meta.get("version", 0))
    iteration = 0
    last = None
    dev = torch.device(device)
    skip = int(meta.get("steps_done", 0)) if resume else 0
    rng_path = os.path.join(checkpoint_dir, f"ps_worker{index}.pt") if checkpoint_dir else None
    if skip and rng_path and os.path.exists(rng_path):
        from ..utils.checkpoint import _set_rng_state
        saved = torch.load(rng_path, map_location="cpu", weights_only=True)
        if int(saved["steps"]) == skip:
            _set_rng_state(saved["rng"], dev)
    wd = None
    if watchdog_s and watchdog_s > 0:
        from ..comm.watchdog import Watchdog
        wd = Watchdog(float(watchdog_s), rank=index)
    try:
        for _ in range(num_epochs):
            for x, y in dataset.iterate(dev):
                if max_steps is not None and iteration >= max_steps:
                    break
                if iteration < skip:           # a step the resumed checkpoint already contains
                    iteration += 1
                    continue
                if fault_hook is not None:
                    fault_hook(index, iteration)
                t0 = time.time()
                last = replica.forward_backward(x, y, None, 1.0)
                replica.finish_backward()
                conn.send(TRAIN, {"version": version, "loss": float(last)}, flat.grad)
                kind, meta, _ = conn.recv(into=host)
                if kind != WEIGHTS:
                    raise RuntimeError(f"expected weights, got message kind {kind}")
                flat.load_flat(host.to(flat.device))
                version = int(meta.get("version", version))
                if iteration % 50 == 0:
                    print(iteration, flush=True)
                if metrics.enabled:
                    metrics.log(step=iteration, loss=float(last), step_s=time.time() - t0, version=version)
                iteration += 1
                if wd is not None:
                    wd.beat(f"step {iteration}")
                if rng_path and meta.get("checkpoint"):
                    from ..utils.checkpoint import _atomic_save, _rng_state
                    os.makedirs(checkpoint_dir, exist_ok=True)
                    _atomic_save({"steps": iteration, "rng": _rng_state(dev)}, rng_path)
        if wd is not None:
            wd.close()
        conn.send(END, {})
    finally:
        conn.close()
        metrics.close()
    res = {"ok": True, "steps": iteration, "loss": float(last) if last is not None else None, "resumed_from": skip}
    if result_queue is not None:
        result_queue.put((f"worker{index}", dict(res)))
    return res
