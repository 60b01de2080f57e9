"""Centralized (parameter-server) strategies with device-resident parameter shards.

The reference parameter server is one process that owns the model, receives
pickled gradients over TCP, applies them under a lock and sends all weights
back (reference centralized/server.py:57-136, centralized/client.py:76-95).
Here the server state -- fp32 master parameters and optimizer moments -- is
*sharded across the GPUs*: rank r owns 1/W of every gradient bucket, sized for
288 GB of HBM per device, and push/pull are collectives over xGMI:

``sync``   push = reduce-scatter of each bucket (started as soon as backward
           has produced it), apply = one fused optimizer launch over the local
           shards, pull = all-gather of the updated parameters.
           ``apply='reduced'`` takes one step on the summed/averaged gradient;
           ``apply='sequential'`` reproduces the reference exactly: every
           worker's gradient is applied separately, in rank order, each with
           its own optimizer state (server.py:52-55,91-93,170).

``async``  no barrier between workers: a worker pushes its gradient shard to
           each owner and receives that owner's current parameters; an owner
           applies each arriving gradient immediately on a side stream, under
           a lock that plays the part of ``model_lock`` (server.py:99-102).
           Each (owner, worker) pair has its own 2-rank communicator and a
           host service thread, because arrival order is data dependent.
           The dedicated topology of the reference (one extra server process,
           ``-n`` workers) is ``placement='dedicated'``.
"""
from __future__ import annotations

import threading
import time

import torch

from ..comm.backend import Comm
from ..engine.optimizer import FlatOptimizer
from ..engine.replica import TrainConfig
from ..obs.trace import phase
from .base import Strategy


class SyncPSStrategy(Strategy):
    graph_capturable = True
    name = "sync"
    replicas_agree = True

    def __init__(self, model, comm: Comm, cfg: TrainConfig | None = None, optimizer=None, metrics=None,
                 apply: str = "reduced", optimizer_state: str | None = None):
        cfg = cfg or TrainConfig()
        cfg.shard_ways = max(cfg.shard_ways, comm.world)
        super().__init__(model, comm, cfg, optimizer, metrics)
        if apply not in ("reduced", "sequential"):
            raise ValueError("apply must be 'reduced' or 'sequential'")
        self.apply = apply
        self.optimizer_state = optimizer_state or ("per_worker" if apply == "sequential" else "shared")
        W, r = comm.world, comm.rank
        self.shards = [self.planner.shard_range(b, r) for b in self.planner.buckets]
        n_states = W if (apply == "sequential" and self.optimizer_state == "per_worker") else 1
        self.optimizers = [FlatOptimizer(self.flat, optimizer, state_ranges=self.shards) for _ in range(n_states)]
        self.optimizer = self.optimizers[0]
        self.broadcast_initial(0)
        self._scale = 1.0 / W if (self.cfg.loss_reduction == "mean" and apply == "reduced") else 1.0
        if apply == "sequential" and W > 1:
            # room for every worker's copy of this rank's shard of every bucket
            self._gathered = torch.empty(self.flat.total, dtype=self.flat.grad.dtype, device=comm.device)
        self.bytes_pushed = 0
        self.bytes_pulled = 0

    # ------------------------------------------------------------------ push
    def _on_ready_reduced(self, b):
        a, e = self.planner.shard_range(b, self.comm.rank)
        self.bytes_pushed += (b.numel - (e - a)) * self.flat.grad.element_size()
        return self.comm.reduce_scatter(self.flat.grad[a:e], self.flat.grad[b.start:b.end], async_op=True)

    def train_step(self, x, y):
        W = self.comm.world
        f = self.flat
        if self.apply == "reduced":
            loss = self.replica.forward_backward(x, y, self._on_ready_reduced if W > 1 else None, self._scale)
            with phase("push_wait"):
                for w in self.replica.finish_backward():
                    if w is not None:
                        w.wait()
            with phase("ps_apply"):
                self.optimizer.step(self.shards)
        else:
            loss = self.replica.forward_backward(x, y, None, 1.0)
            self.replica.finish_backward()
            if W == 1:
                self.optimizer.step(self.shards)
            else:
                # every owner receives each worker's gradient for its shards, then applies them one
                # worker after the other (rank order), as the reference's per-connection threads do
                per_worker = [dict() for _ in range(W)]
                for b, (a, e) in zip(self.planner.buckets, self.shards):
                    n = e - a
                    g = self._gathered[b.start:b.end]
                    self.comm.all_to_all(g, f.grad[b.start:b.end])
                    self.bytes_pushed += (W - 1) * n * f.grad.element_size()
                    for w in range(W):
                        per_worker[w][(a, e)] = g[w * n:(w + 1) * n]
                for w in range(W):
                    opt = self.optimizers[w if len(self.optimizers) > 1 else 0]
                    opt.step(self.shards, grad=_Mapped(per_worker[w]))
        # ------------------------------------------------------------- pull
        if W > 1:
            with phase("pull"):
                for b, (a, e) in zip(self.planner.buckets, self.shards):
                    self.comm.all_gather(f.master[b.start:b.end], f.master[a:e])
                    self.bytes_pulled += (b.numel - (e - a)) * 4
                f.refresh_shadow()
        self.apply_sparse()
        return loss


class _Mapped:
    """Gradient source for ``FlatOptimizer.step``: arena range -> the tensor holding it."""

    def __init__(self, table: dict):
        self.table = table

    def __getitem__(self, s: slice):
        return self.table[(s.start, s.stop)]


class _Shifted:
    """Indexes ``base`` with arena coordinates: ``self[a:b] == base[a + shift : b + shift]``."""

    def __init__(self, base: torch.Tensor, shift: int):
        self.base, self.shift = base, shift

    def __getitem__(self, s: slice):
        return self.base[s.start + self.shift:s.stop + self.shift]


# ============================================================================ async
_CMD_TRAIN, _CMD_END = 1, 2


class AsyncPSStrategy(Strategy):
    """Asynchronous parameter server over per-pair communicators.

    ``placement='colocated'``: every rank is a worker and owns shard r of the arena.
    ``placement='dedicated'``: rank 0 is the server (owns everything, does not train), ranks
    1..W-1 are the workers -- the process layout of the reference launcher
    (reference initializer.py:134-145).
    """
    name = "async"

    def __init__(self, model, comm: Comm, cfg: TrainConfig | None = None, optimizer=None, metrics=None,
                 placement: str = "colocated", max_staleness: int | None = None):
        cfg = cfg or TrainConfig()
        if placement not in ("colocated", "dedicated"):
            raise ValueError("placement must be 'colocated' or 'dedicated'")
        W = comm.world
        self.placement = placement if W > 1 else "colocated"
        self.servers = list(range(W)) if self.placement == "colocated" else [0]
        self.workers = list(range(W)) if self.placement == "colocated" else list(range(1, W))
        cfg.shard_ways = max(cfg.shard_ways, len(self.servers))
        super().__init__(model, comm, cfg, optimizer, metrics)
        r = comm.rank
        S = len(self.servers)
        n = self.flat.total // S
        self.shard_of = {s: (i * n, (i + 1) * n) for i, s in enumerate(self.servers)}
        self.is_server = r in self.servers
        self.is_worker = r in self.workers
        self.max_staleness = max_staleness
        self.broadcast_initial(0)
        # one 2-rank communicator per ordered (server, worker) pair; creation is collective
        self.pair: dict[tuple[int, int], Comm] = {}
        if W > 1:
            for s in self.servers:
                for w in self.workers:
                    if s != w:
                        c = comm.subgroup([s, w])
                        if r in (s, w):
                            self.pair[(s, w)] = c
        self.lock = threading.Lock()          # the reference's model_lock
        self.version = 0                      # number of gradient applications on this shard
        self.staleness_log: list[int] = []
        self._threads: list[threading.Thread] = []
        self._errors: list[BaseException] = []
        if self.is_server:
            a, e = self.shard_of[r]
            self.optimizer = FlatOptimizer(self.flat, optimizer, state_ranges=[(a, e)])
            self._apply_stream = self.side_stream()
            for w in self.workers:
                if w != r:
                    t = threading.Thread(target=self._serve, args=(w,), daemon=True, name=f"ps-serve-{w}")
                    t.start()
                    self._threads.append(t)
        self._versions_seen = {s: 0 for s in self.servers}

    # ---------------------------------------------------------------- server
    def _apply_locked(self, grad_shard: torch.Tensor, version_used: int) -> int:
        a, e = self.shard_of[self.comm.rank]
        with self.lock:
            self.staleness_log.append(self.version - version_used)
            self.optimizer.step([(a, e)], grad=_Shifted(grad_shard, -a))
            self.version += 1
            return self.version

    def _serve(self, worker: int) -> None:
        """Service thread for one worker: receive (command, gradient shard), apply, reply with the
        current parameter shard -- the body of the reference's ``Server.run`` loop."""
        try:
            r = self.comm.rank
            c = self.pair[(r, worker)]
            peer = 1 - c.rank
            a, e = self.shard_of[r]
            dev = self.comm.device
            ctrl = torch.zeros(2, dtype=torch.int64, device=dev)
            grad = torch.empty(e - a, dtype=self.flat.grad.dtype, device=dev)
            reply = torch.empty(e - a, dtype=torch.float32, device=dev)
            stream = torch.cuda.Stream(dev) if dev.type == "cuda" else None
            while True:
                with (torch.cuda.stream(stream) if stream is not None else _null()):
                    c.recv(ctrl, peer)
                    cmd, used = (int(v) for v in ctrl.tolist())
                    if cmd == _CMD_END:
                        break
                    c.recv(grad, peer)
                    ver = self._apply_locked(grad, used)
                    with self.lock:
                        reply.copy_(self.flat.master[a:e])
                    ctrl[0], ctrl[1] = _CMD_TRAIN, ver
                    c.send(ctrl, peer)
                    c.send(reply, peer)
        except BaseException as ex:  # surfaced by join_servers()
            self._errors.append(ex)

    # ---------------------------------------------------------------- worker
    def train_step(self, x, y):
        if not self.is_worker:
            raise RuntimeError("a dedicated server rank does not train; call serve_until_done()")
        f = self.flat
        r = self.comm.rank
        loss = self.replica.forward_backward(x, y, None, 1.0)
        self.replica.finish_backward()
        dev = self.comm.device
        for s in self.servers:
            a, e = self.shard_of[s]
            if s == r:
                ver = self._apply_locked(f.grad[a:e], self._versions_seen[s])
                self._versions_seen[s] = ver
                continue
            c = self.pair[(s, r)]
            peer = 1 - c.rank
            ctrl = torch.tensor([_CMD_TRAIN, self._versions_seen[s]], dtype=torch.int64, device=dev)
            c.send(ctrl, peer)
            c.send(f.grad[a:e], peer)
            c.recv(ctrl, peer)
            c.recv(f.master[a:e], peer)
            self._versions_seen[s] = int(ctrl[1])
        f.refresh_shadow()
        return loss

    def finalize(self) -> None:
        r = self.comm.rank
        dev = self.comm.device
        if self.is_worker:
            for s in self.servers:
                if s != r:
                    c = self.pair[(s, r)]
                    c.send(torch.tensor([_CMD_END, 0], dtype=torch.int64, device=dev), 1 - c.rank)
        self.join_servers()
        # final model = the servers' shards
        if self.comm.world > 1:
            for s in self.servers:
                a, e = self.shard_of[s]
                self.comm.broadcast(self.flat.master[a:e], s)
            self.flat.refresh_shadow()

    def join_servers(self, timeout: float = 600.0) -> None:
        for t in self._threads:
            t.join(timeout)
            if t.is_alive():
                raise TimeoutError(f"parameter-server thread {t.name} did not finish")
        if self._errors:
            raise RuntimeError(f"parameter-server thread failed: {self._errors[0]!r}") from self._errors[0]

    def serve_until_done(self) -> None:
        """Body of a dedicated server rank: wait until every worker has sent ``end``."""
        self.join_servers()

    def run_epochs(self, dataset, epochs: int = 1, verbose: int = 0, max_steps: int | None = None, fault_hook=None):
        if self.is_worker:
            # async workers do not wait for each other between steps: only start and end are collective
            n_local = dataset.num_batches()
            steps = n_local if max_steps is None else min(n_local, max_steps)
            self.sync_device()
            self.comm.barrier()
            t0 = time.time()
            it_count, last = 0, None
            for ep in range(epochs):
                it = dataset.iterate(self.comm.device)
                for _ in range(steps):
                    x, y = next(it)
                    if fault_hook is not None:
                        fault_hook(self.comm.rank, it_count)
                    last = self.train_step(x, y)
                    if verbose and it_count % 50 == 0:
                        print(it_count, flush=True)
                    it_count += 1
            self.finalize()
            self.sync_device()
            self.comm.barrier()
            return {"elapsed": time.time() - t0, "steps": it_count, "loss": float(last) if last is not None else None,
                    "mean_staleness": (sum(self.staleness_log) / max(1, len(self.staleness_log)))}
        self.comm.barrier()
        t0 = time.time()
        self.finalize()
        self.comm.barrier()
        return {"elapsed": time.time() - t0, "steps": 0, "loss": None,
                "mean_staleness": (sum(self.staleness_log) / max(1, len(self.staleness_log)))}


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
