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
           applies each arriving gradient immediately (server.py:99-102). See
           :class:`AsyncPSStrategy` for the stream / event / thread design.
           The dedicated topology of the reference (one extra server process,
           ``-n`` workers) is ``placement='dedicated'``.
"""
from __future__ import annotations

import os
import socket
import threading
import time

import torch

from ..comm.backend import Comm
from ..engine.optimizer import FlatOptimizer
from ..engine.replica import TrainConfig
from ..obs.trace import phase
from .base import Strategy


class SyncPSStrategy(Strategy):
    """Synchronous parameter server whose state is SHARDED across the ranks.

    Rank r is the server of shard r of every gradient bucket. What it keeps of the server's state -- the fp32
    master copy of the bf16 parameters and the optimizer moments -- is that shard only (1/W of the model; the
    reference server owns the one model, reference centralized/server.py:148-150). Every rank keeps the working
    parameters whole: the bf16 arena the layers read, and the fp32 parameters (norm scales / offsets, a model
    trained in fp32), which are their own master copy.

    Per bucket and step: reduce-scatter of the gradients while backward still runs (push) -> ONE fused optimizer
    launch on the owned shard that also writes the shard's bf16 working copy -> all-gather of that bf16 arena slice
    (pull; fp32 parameters travel as fp32). The all-gathers are posted back to back, first-needed bucket first, and
    nobody waits for them at the end of the step: the NEXT forward pass waits for a bucket's pull when the first
    layer holding one of its parameters is about to run (``pull_wait='first_use'``), so the bulk of the pull -- the
    late layers' parameters -- travels under the early layers' forward. ``pull_wait='step_end'`` waits before the
    step returns (the form a captured hipGraph needs, and the safe choice for a model that reads a parameter
    outside the module that owns it, before that module has run).
    """
    graph_capturable = True
    name = "sync"
    replicas_agree = True

    def __init__(self, model, comm: Comm, cfg: TrainConfig | None = None, optimizer=None, metrics=None,
                 apply: str = "reduced", optimizer_state: str | None = None, pull_wait: str | None = None,
                 shard_master: bool = True):
        cfg = cfg or TrainConfig()
        cfg.shard_ways = max(cfg.shard_ways, comm.world)
        cfg.split_dtypes = True              # bf16 parameters first: below the boundary the fp32 master is sharded
        super().__init__(model, comm, cfg, optimizer, metrics)
        if apply not in ("reduced", "sequential"):
            raise ValueError("apply must be 'reduced' or 'sequential'")
        if pull_wait not in (None, "first_use", "step_end"):
            raise ValueError("pull_wait must be 'first_use' or 'step_end'")
        self.apply = apply
        self.optimizer_state = optimizer_state or ("per_worker" if apply == "sequential" else "shared")
        W, r = comm.world, comm.rank
        self.shards = [self.planner.shard_range(b, r) for b in self.planner.buckets]
        n_states = W if (apply == "sequential" and self.optimizer_state == "per_worker") else 1
        self.optimizers = [FlatOptimizer(self.flat, optimizer, state_ranges=self.shards) for _ in range(n_states)]
        self.optimizer = self.optimizers[0]
        self.broadcast_initial(0)
        if shard_master and self.flat.split:
            # from here on this rank holds 1/W of the fp32 master of the bf16 parameters
            self.flat.shard_master([(a, e) for a, e in self.shards if a < self.flat.split])
        self._scale = 1.0 / W if (self.cfg.loss_reduction == "mean" and apply == "reduced") else 1.0
        if apply == "sequential" and W > 1:
            # room for every worker's copy of this rank's shard of every bucket
            self._gathered = torch.empty(self.flat.total, dtype=self.flat.grad.dtype, device=comm.device)
        self.bytes_pushed = 0
        self.bytes_pulled = 0
        self.pull_wait = pull_wait or "first_use"
        self._pulls: dict[int, object] = {}        # bucket index -> work handle of its all-gather in flight
        self._hooks = []
        if W > 1:
            self._install_pull_hooks()

    # ------------------------------------------------------------------ pull
    def _install_pull_hooks(self) -> None:
        """A forward pre-hook on every module that owns parameters: before it runs, the pulls of the buckets its
        parameters live in must have landed."""
        for mod in self.replica.model.modules():
            idx = sorted({self.planner.slot_bucket[id(p)].index for p in mod._parameters.values()
                          if p is not None and id(p) in self.planner.slot_bucket})
            if idx:
                self._hooks.append(mod.register_forward_pre_hook(self._make_hook(tuple(idx))))

    def _make_hook(self, buckets: tuple):
        pulls = self._pulls

        def hook(_mod, _inp):
            if pulls:
                for i in buckets:
                    w = pulls.pop(i, None)
                    if w is not None:
                        w.wait()
        return hook

    def wait_pulls(self) -> None:
        """Every pull in flight has landed (before anything but the next forward pass reads the parameters)."""
        for i in list(self._pulls):
            w = self._pulls.pop(i)
            if w is not None:
                w.wait()

    def _pull_view(self, a: int, e: int) -> torch.Tensor:
        """What an owner broadcasts of [a, e): with a sharded master the bf16 working copy of bf16 parameters (2
        bytes a parameter: nobody else holds their fp32 master), otherwise the fp32 master -- which for fp32
        parameters is the parameter itself."""
        f = self.flat
        if f.master_base and a < f.master_base:
            return f.shadow[a:e]
        return f.m32(a, e)

    def _post_pulls(self) -> None:
        f = self.flat
        # last bucket first: it holds the first layers, the ones the next forward pass needs first
        for b, (a, e) in reversed(list(zip(self.planner.buckets, self.shards))):
            whole, mine = self._pull_view(b.start, b.end), self._pull_view(a, e)
            self._pulls[b.index] = self.comm.all_gather(whole, mine, async_op=True)
            self.bytes_pulled += (b.numel - (e - a)) * whole.element_size()
        if f.shadow is not None and not f.master_base:
            self.wait_pulls()
            f.refresh_shadow()       # whole fp32 master on every rank: the bf16 copies follow the gathered values
        elif self.pull_wait == "step_end" or self.cfg.graph:
            self.wait_pulls()

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
        if W > 1:
            with phase("pull"):
                self._post_pulls()
        self.apply_sparse()
        return loss

    # ------------------------------------------------------- whole-model views
    def evaluate(self, dataset, reduce: bool = True, max_batches: int | None = None):
        self.wait_pulls()
        return super().evaluate(dataset, reduce, max_batches)

    def check_replicas(self, step: int = -1) -> None:
        self.wait_pulls()
        super().check_replicas(step)

    def finalize(self) -> None:
        self.wait_pulls()

    def export_master(self) -> torch.Tensor:
        """The whole fp32 master on every rank (collective): the owners' shards all-gathered into a temporary."""
        self.wait_pulls()
        f = self.flat
        if not f.master_base:
            return f.master.detach().clone()
        full = torch.empty(f.total, dtype=torch.float32, device=f.device)
        full[f.master_base:].copy_(f.master)
        for b, (a, e) in zip(self.planner.buckets, self.shards):
            if a < f.split:
                full[a:e].copy_(f.m32(a, e))
                w = self.comm.all_gather(full[b.start:b.end], full[a:e])
                if w is not None:
                    w.wait()
        return full

    def import_master(self, full: torch.Tensor) -> None:
        self.wait_pulls()
        self.flat.write_master(full)

    def on_weights_loaded(self) -> None:
        """A weight import wrote the working parameters (bf16 copies below the boundary): the owned fp32 shards
        follow them, so that the next optimizer step starts from the imported model."""
        f = self.flat
        if f.master_base:
            ranges, store = f.master_shards
            base = 0
            with torch.no_grad():
                for a, e in ranges:
                    store[base:base + e - a].copy_(f.shadow[a:e].float())
                    base += e - a


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
_K_HELLO, _K_TRAIN, _K_REPLY, _K_END = 1, 2, 3, 4

# Host-side serialisation of RCCL calls. A process drives several two-rank communicators here (one per
# owner/worker pair) from its training thread and its service threads; RCCL gives no ordering guarantee for
# calls that race on different communicators of one process, so every call is made under this lock: from
# RCCL's point of view one thread issues them all, one after the other. A call only ENQUEUES work (kernels on
# the communicator's stream; completion is a stream dependency, never a host wait), so the lock is held for
# microseconds; the connections are set up beforehand, in one global order, where the first call on a
# communicator does block on its peer. gloo calls block until they are matched and must not share a lock.
_RCCL_CALLS = threading.Lock()


class _NoLock:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _host_address(master_addr: str | None) -> str:
    """The address other ranks reach this one at (control connections of the asynchronous server)."""
    env = os.environ.get("DTF_HOST_ADDR")
    if env:
        return env
    if master_addr in (None, "", "127.0.0.1", "localhost"):
        return "127.0.0.1"
    try:
        return socket.gethostbyname(socket.gethostname())
    except OSError:
        return "127.0.0.1"


class AsyncPSStrategy(Strategy):
    """Asynchronous parameter server with device-resident shards.

    ``placement='colocated'``: every rank is a worker and owns shard r of the arena.
    ``placement='dedicated'``: rank 0 is the server (owns everything, does not train), ranks
    1..W-1 are the workers -- the process layout of the reference launcher
    (reference initializer.py:134-145).

    Roles are kept apart as in the reference, where the server owns the model and every client a copy
    (reference centralized/server.py:148-150, client.py:65,72): an owner's authoritative shard (``ps_master``
    and the optimizer moments) is separate memory from the replica's working parameters (``flat.master``), which
    only the training thread writes -- by its pulls.

    Control plane: one framed-TCP connection per (owner, worker) pair and, on the owner, one host thread per
    worker blocked in a socket read -- the reference's thread-per-connection server (server.py:168-174). A push
    is ANNOUNCED there (``train``, the parameter version the gradient was computed on) and answered there (the
    version after the apply): versions and staleness are host integers, and no thread ever parks a receive
    kernel on the GPU or reads a device tensor back to learn what to do next.

    Data plane, ``transport``:

    ``'ipc'`` (GPUs of one host; the default there): PULL ONLY over HIP-IPC peer pointers, and no device kernel ever
    waits for anything outside its own GPU. Every rank maps every other rank's gradient arena and reply buffer. A
    worker makes sure its gradient is in memory (one host wait on an event), then announces the push; the owner's
    service thread launches the ordinary fused optimizer kernel with the worker's gradient shard -- read straight
    through the peer pointer, over the pair's xGMI link -- as its gradient operand, copies the updated shard into the
    reply slot it keeps for that worker, waits (host) for that copy and answers; the worker then copies the slot into
    its own arena through the peer pointer. Nobody writes into another GPU's memory; a kernel that reads peer memory
    is preceded by a system-scope acquire on every XCD (``dtf_p2p_acquire``), so no cache holds an older copy.
    Both hand-offs are host-synchronous (an event wait plus a TCP frame: tens of microseconds against a step of
    milliseconds), which is what makes the plane deadlock-free BY CONSTRUCTION: the only waits are host threads
    waiting for their own device or for a frame, and a frame is always sent by a thread that waits for nothing
    remote.

    ``'p2p'``: one two-rank communicator per ordered (owner, worker) pair (RCCL send/recv; gloo on CPU), SENDS
    FIRST: a worker posts ``send(gradient shard)`` to every owner (ascending rank) and only then announces; the
    owner posts the matching receive when the announcement arrives, applies, posts ``send(parameters)`` and only then
    answers; the worker posts the matching receive when the answer arrives (ascending rank). A transfer's second
    half is therefore always posted by a host thread that is waiting for nothing on a device, and no kernel spans
    another process's apply. A send / receive kernel still parks on its GPU until its partner runs, so on RCCL the
    plane is only as safe as the hardware queues are independent: with ``placement='dedicated'`` every worker has at
    most one transfer outstanding and the server's transfers each have their partner at the head of a worker's
    queue -- no wait-for cycle exists whatever stream shares a hardware queue with whatever other. Colocated
    owners on RCCL have both roles' kernels in the at most four hardware queues of a process, where a ring of
    parked kernels (rank i's send behind rank i's receive) cannot be excluded: that combination is refused, the
    IPC plane (one host) or the dedicated placement is what to use.

    ``model_lock`` (server.py:99-102) is an event chain on ONE apply stream per owner. Every apply -- a remote
    worker's and, colocated, the owner's own -- is enqueued on that stream, under a host lock that only fixes
    the enqueue order: receive of the gradient (the stream waits for its arrival) -> fused optimizer kernel on
    the owner's shard -> copy of the updated shard into that worker's reply buffer -> send (which waits for the
    copy). Two applies can therefore never overlap on the device, a reply is a snapshot between two applies, and
    nothing is lost; the training stream joins the chain through a "gradient ready" event before its own apply
    and an "applied" event after it.
    """
    name = "async"

    def __init__(self, model, comm: Comm, cfg: TrainConfig | None = None, optimizer=None, metrics=None,
                 placement: str = "colocated", max_staleness: int | None = None, master_addr: str | None = None,
                 transport: str = "auto", staleness_lr: bool | None = None):
        cfg = cfg or TrainConfig()
        if placement not in ("colocated", "dedicated"):
            raise ValueError("placement must be 'colocated' or 'dedicated'")
        if transport not in ("auto", "ipc", "p2p"):
            raise ValueError("transport must be 'auto', 'ipc' or 'p2p'")
        W = comm.world
        self.placement = placement if W > 1 else "colocated"
        self.servers = list(range(W)) if self.placement == "colocated" else [0]
        self.workers = list(range(W)) if self.placement == "colocated" else list(range(1, W))
        cfg.shard_ways = max(cfg.shard_ways, len(self.servers))
        super().__init__(model, comm, cfg, optimizer, metrics)
        r = comm.rank
        dev = comm.device
        S = len(self.servers)
        n = self.flat.total // S
        self.shard_of = {s: (i * n, (i + 1) * n) for i, s in enumerate(self.servers)}
        self.is_server = r in self.servers
        self.is_worker = r in self.workers
        self.max_staleness = max_staleness
        # Staleness-aware step size: a gradient computed on parameters that are k applies old is applied with
        # lr / (1 + k). With N workers in step every gradient is N - 1 applies old, so a round of N applies moves
        # the parameters like ONE step on the averaged gradient; without it (the reference's behaviour, server.py:
        # 98-102: every gradient at the full rate) a round is N full steps and an 8-worker run on a task with label
        # noise ends 50 points below all-reduce (profiles/convergence_strategies_v2.txt). DTF_ASYNC_STALENESS_LR=0
        # or staleness_lr=False restores the reference's rule.
        if staleness_lr is None:
            staleness_lr = os.environ.get("DTF_ASYNC_STALENESS_LR", "1") != "0"
        self.staleness_lr = bool(staleness_lr)
        self.broadcast_initial(0)
        self.lock = threading.Lock()          # the reference's model_lock: fixes the enqueue order of applies
        self.version = 0                      # gradient applications enqueued on this shard so far
        self.staleness_log: list[int] = []
        self._threads: list[threading.Thread] = []
        self._errors: list[BaseException] = []
        self._versions_seen = {s: 0 for s in self.servers}
        self._apply_stream = torch.cuda.Stream(dev) if dev.type == "cuda" else None
        self._calls = _RCCL_CALLS if comm.backend == "nccl" else _NoLock()
        self.pair: dict[tuple[int, int], Comm] = {}
        self._ctrl: dict[int, object] = {}    # worker side: owner rank -> control connection
        self.transport = self._pick_transport(transport) if W > 1 else "p2p"
        self._peers = None                    # IPC plane: comm.ipc.PeerGroup
        self._grad_of: dict[int, torch.Tensor] = {}    # rank -> its gradient arena, through the peer pointer
        self._reply_of: dict[int, torch.Tensor] = {}   # rank -> its reply buffer, through the peer pointer
        self._reply_all = None
        self.ps_master = None
        if self.is_server:
            a, e = self.shard_of[r]
            self.ps_master = self.flat.master[a:e].clone()     # the authoritative shard
            self.optimizer = FlatOptimizer(self.flat, optimizer, state_ranges=[(a, e)])
            self._base_lr = float(self.optimizer.spec.learning_rate)
        if W > 1:
            self._connect(master_addr)

    # ---------------------------------------------------------------- set-up
    def _pick_transport(self, want: str) -> str:
        comm = self.comm
        gpu = comm.device.type == "cuda"
        if want == "ipc" and not gpu:
            raise ValueError("transport='ipc' needs GPU devices")
        if want == "auto":
            want = "p2p"
            if gpu:
                import torch.distributed as dist
                names = [None] * comm.world
                dist.all_gather_object(names, socket.gethostname(), group=comm.group)
                if len(set(names)) == 1 and os.environ.get("DTF_ASYNC_TRANSPORT", "ipc") == "ipc":
                    want = "ipc"
        if want == "p2p" and comm.backend == "nccl" and self.placement == "colocated":
            raise ValueError("asynchronous parameter server: colocated shard owners over RCCL send/recv cannot be shown "
                             "free of wait-for cycles between parked kernels; use the IPC plane (ranks on one host), "
                             "--placement dedicated, or the TCP transport")
        return want

    def _peer_view(self, ptr: int, numel: int, dtype: torch.dtype) -> torch.Tensor:
        """A tensor over ``numel`` elements at the raw device address ``ptr`` (peer memory mapped by HIP IPC)."""
        esz = 4 if dtype == torch.float32 else 2
        spec = {"shape": (numel,), "typestr": "<f4" if dtype == torch.float32 else "<u2", "data": (int(ptr), False),
                "version": 2, "strides": None}
        holder = type("_PeerMemory", (), {"__cuda_array_interface__": spec})()
        t = torch.as_tensor(holder, device=self.comm.device)
        if dtype == torch.bfloat16:
            t = t.view(torch.bfloat16)
        assert t.data_ptr() == ptr and t.numel() * esz == numel * esz
        return t

    def _connect_ipc(self) -> None:
        """Maps every rank's gradient arena and reply buffer into this process (collective)."""
        from ..comm.ipc import PeerGroup
        comm, r, f = self.comm, self.comm.rank, self.flat
        self._peers = PeerGroup(comm)
        n = f.total // len(self.servers)
        # reply slot of worker w at an owner: [w * n, (w + 1) * n) of the owner's buffer (fp32 shard snapshots)
        self._reply_all = torch.zeros(comm.world * n if self.is_server else 64, dtype=torch.float32, device=comm.device)
        kg = self._peers.register(f.grad)
        kr = self._peers.register(self._reply_all)
        gp, rp = self._peers._registered[kg][0], self._peers._registered[kr][0]
        for j in range(comm.world):
            self._grad_of[j] = f.grad if j == r else self._peer_view(gp[j], f.total, f.grad.dtype)
            if j in self.servers:
                self._reply_of[j] = self._reply_all if j == r else self._peer_view(rp[j], comm.world * n, torch.float32)

    def _acquire(self) -> None:
        """Before a kernel that reads peer memory: every XCD drops its cached copies of memory it does not own."""
        from ..ops import _lib
        _lib.check(_lib.kernels().dtf_p2p_acquire(_lib.stream_ptr(self.comm.device)), "p2p acquire")

    def _connect(self, master_addr) -> None:
        from ..comm import tcp
        comm, r, dev = self.comm, self.comm.rank, self.comm.device
        W = comm.world
        pairs = [(s, w) for s in self.servers for w in self.workers if s != w]
        if self.transport == "ipc":
            self._connect_ipc()
            pairs = []
        # data plane: creation is collective on the parent group, in one order on every rank
        for s, w in pairs:
            c = comm.subgroup([s, w])
            if r in (s, w):
                self.pair[(s, w)] = c
        # control plane: owners listen on an ephemeral port; (address, port) travel by all-gather
        listener = None
        port = 0
        if self.is_server and any(w != r for w in self.workers):
            addr = _host_address(master_addr)
            listener = tcp.Listener("127.0.0.1" if addr == "127.0.0.1" else "0.0.0.0", 0)
            port = listener.port
        ip = [int(v) for v in _host_address(master_addr).split(".")]
        mine = torch.tensor(ip + [port], dtype=torch.int64, device=dev)
        table = torch.zeros(W * 5, dtype=torch.int64, device=dev)
        comm.all_gather(table, mine)
        table = table.cpu().reshape(W, 5).tolist()
        if self.is_worker:
            for s in self.servers:
                if s != r:
                    host = ".".join(str(v) for v in table[s][:4])
                    c = tcp.connect(host, table[s][4], timeout=120.0, conn_timeout=600.0)
                    c.send(_K_HELLO, {"rank": r})
                    self._ctrl[s] = c
        conns = {}
        if listener is not None:
            for _ in [w for w in self.workers if w != r]:
                c = listener.accept(timeout=120.0, conn_timeout=-1.0)
                kind, meta, _ = c.recv(timeout=120.0)
                if kind != _K_HELLO:
                    raise RuntimeError("asynchronous parameter server: unexpected first control message")
                conns[int(meta["rank"])] = c
            listener.close()
        # first use of a communicator sets its connection up and blocks on the peer: do it here, pair by pair in
        # the global order, so that later calls only enqueue
        a0 = torch.zeros(8, dtype=torch.float32, device=dev)
        for s, w in pairs:
            if r == s:
                c = self.pair[(s, w)]
                c.recv(a0, 1 - c.rank)
                c.send(a0, 1 - c.rank)
            elif r == w:
                c = self.pair[(s, w)]
                c.send(a0, 1 - c.rank)
                c.recv(a0, 1 - c.rank)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        for w, c in conns.items():
            t = threading.Thread(target=self._serve, args=(w, c), daemon=True, name=f"ps-serve-{w}")
            t.start()
            self._threads.append(t)

    # ---------------------------------------------------------------- server
    def _stream_ctx(self):
        return torch.cuda.stream(self._apply_stream) if self._apply_stream is not None else _NoLock()

    def _apply(self, grad_shard: torch.Tensor, version_used: int) -> int:
        """One optimizer step on the authoritative shard; the caller holds ``self.lock`` and has made the apply
        stream current."""
        a, e = self.shard_of[self.comm.rank]
        stale = max(0, self.version - version_used)
        self.staleness_log.append(stale)
        if self.staleness_lr:
            # (a device-resident scalar: the write is ordered on the apply stream like the kernel that reads it)
            self.optimizer.set_lr(self._base_lr / (1.0 + stale))
        self.optimizer.step([(a, e)], grad=_Shifted(grad_shard, -a), params=_Shifted(self.ps_master, -a))
        self.version += 1
        return self.version

    def _serve(self, worker: int, conn) -> None:
        """Service thread of one worker: wait for an announcement on the control connection, then enqueue
        receive -> apply -> reply on the apply stream -- the body of the reference's ``Server.run`` loop
        (reference centralized/server.py:62-105)."""
        try:
            if self.transport == "ipc":
                return self._serve_ipc(worker, conn)
            r = self.comm.rank
            c = self.pair[(r, worker)]
            peer = 1 - c.rank
            a, e = self.shard_of[r]
            dev = self.comm.device
            grad = torch.empty(e - a, dtype=self.flat.grad.dtype, device=dev)
            reply = torch.empty(e - a, dtype=torch.float32, device=dev)
            send_stream = torch.cuda.Stream(dev) if dev.type == "cuda" else None
            while True:
                kind, meta, _ = conn.recv()
                if kind == _K_END:
                    break
                if kind != _K_TRAIN:
                    raise RuntimeError(f"asynchronous parameter server: unknown control message {kind}")
                copied = None
                with self.lock, self._stream_ctx():
                    # the worker posted its send before it announced the push: the receive is the transfer's second
                    # half and completes without anybody's further help
                    with self._calls:
                        works = c.exchange([], [(grad, peer)])
                    for w_ in works:
                        w_.wait()
                    ver = self._apply(grad, int(meta["used"]))
                    reply.copy_(self.ps_master)
                    if send_stream is not None:
                        copied = torch.cuda.Event()
                        copied.record()
                # the reply leaves outside the lock (its buffer belongs to this thread): a worker that is slow to
                # take it delays nobody else's apply. The send is posted BEFORE the answer: the worker posts the
                # matching receive when the answer arrives
                with (torch.cuda.stream(send_stream) if send_stream is not None else _NoLock()):
                    if copied is not None:
                        send_stream.wait_event(copied)
                    with self._calls:
                        works = c.exchange([(reply, peer)], [])
                conn.send(_K_REPLY, {"version": ver})
                for w_ in works:
                    w_.wait()
            conn.close()
        except BaseException as ex:  # surfaced by join_servers()
            self._errors.append(ex)

    def _serve_ipc(self, worker: int, conn) -> None:
        """IPC plane: the optimizer kernel reads the worker's gradient shard through the peer pointer; the updated
        shard is copied into this worker's reply slot; the answer leaves when that copy is in memory."""
        r = self.comm.rank
        a, e = self.shard_of[r]
        n = e - a
        slot = self._reply_all[worker * n:(worker + 1) * n]
        grad = self._grad_of[worker][a:e]
        while True:
            kind, meta, _ = conn.recv()
            if kind == _K_END:
                break
            if kind != _K_TRAIN:
                raise RuntimeError(f"asynchronous parameter server: unknown control message {kind}")
            with self.lock, self._stream_ctx():
                self._acquire()
                ver = self._apply(grad, int(meta["used"]))
                slot.copy_(self.ps_master)
                done = torch.cuda.Event()
                done.record()
            done.synchronize()            # outside the lock: the next apply is already being enqueued
            conn.send(_K_REPLY, {"version": ver})
        conn.close()

    # ---------------------------------------------------------------- worker
    def train_step(self, x, y):
        if not self.is_worker:
            raise RuntimeError("a dedicated server rank does not train; call serve_until_done()")
        loss = self.replica.forward_backward(x, y, None, 1.0)
        self.replica.finish_backward()
        with phase("push_pull"):
            self.push_pull()
        return loss

    def push_pull(self) -> None:
        """Pushes ``flat.grad`` to the shard owners and pulls their parameters into ``flat.master`` -- one
        iteration of the reference worker's send / receive / ``set_weights`` (reference client.py:85-90)."""
        f = self.flat
        r = self.comm.rank
        dev = self.comm.device
        remote = sorted(s for s in self.servers if s != r)
        ipc = self.transport == "ipc"
        works = []
        if ipc:
            # the owners read the gradient where it lies: it must BE in memory before the announcement leaves
            ready_host = torch.cuda.Event()
            ready_host.record()
            ready_host.synchronize()
        else:
            # first halves: one send per owner, ascending rank, all posted before anything is announced, so that all
            # owners -- and all links -- work at once
            for s in remote:
                a, e = self.shard_of[s]
                c = self.pair[(s, r)]
                with self._calls:
                    works += c.exchange([(f.grad[a:e], 1 - c.rank)], [])
        for s in remote:
            self._ctrl[s].send(_K_TRAIN, {"used": self._versions_seen[s]})
        if r in self.shard_of:
            # colocated: this rank's own shard goes through the same chain, without a transfer
            a, e = self.shard_of[r]
            ready = applied = None
            if self._apply_stream is not None:
                ready = torch.cuda.Event()
                ready.record()
            with self.lock, self._stream_ctx():
                if ready is not None:
                    self._apply_stream.wait_event(ready)
                self._versions_seen[r] = self._apply(f.grad[a:e], self._versions_seen[r])
                f.master[a:e].copy_(self.ps_master)
                if ready is not None:
                    applied = torch.cuda.Event()
                    applied.record()
            if applied is not None:
                torch.cuda.current_stream(dev).wait_event(applied)
        for s in remote:                      # ascending rank: the order the receives are posted in is fixed
            kind, meta, _ = self._ctrl[s].recv()
            if kind != _K_REPLY:
                raise RuntimeError("asynchronous parameter server: unexpected reply on the control connection")
            self._versions_seen[s] = int(meta["version"])
            a, e = self.shard_of[s]
            if ipc:
                # the owner's copy into this worker's slot is in memory (it waited for it): pull it
                n = e - a
                self._acquire()
                f.master[a:e].copy_(self._reply_of[s][r * n:(r + 1) * n])
            else:
                c = self.pair[(s, r)]
                with self._calls:
                    works += c.exchange([], [(f.master[a:e], 1 - c.rank)])
        for w_ in works:
            w_.wait()
        f.refresh_shadow()

    def on_weights_loaded(self) -> None:
        """The authoritative shard follows an import / resume that rewrote the replica's arena; otherwise the
        first apply would continue from the weights the server was constructed with."""
        if self.ps_master is not None:
            a, e = self.shard_of[self.comm.rank]
            if self._apply_stream is not None:
                self._apply_stream.wait_stream(torch.cuda.current_stream(self.comm.device))
            with self.lock, self._stream_ctx():
                self.ps_master.copy_(self.flat.master[a:e])
            if self._apply_stream is not None:
                torch.cuda.current_stream(self.comm.device).wait_stream(self._apply_stream)

    def finalize(self) -> None:
        r = self.comm.rank
        if self.is_worker:
            for s, c in self._ctrl.items():
                c.send(_K_END, {})
                c.close()
            self._ctrl = {}
        self.join_servers()
        if self._peers is not None:
            self.comm.barrier()            # nobody unmaps a buffer that a peer's last pull may still read
            self._grad_of, self._reply_of = {}, {}
            self._peers.close()
            self._peers = None
        # final model = the owners' shards
        if self.ps_master is not None:
            a, e = self.shard_of[r]
            if self._apply_stream is not None:
                torch.cuda.current_stream(self.comm.device).wait_stream(self._apply_stream)
            self.flat.master[a:e].copy_(self.ps_master)
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
        self._threads = []
        if self._errors:
            raise RuntimeError(f"parameter-server thread failed: {self._errors[0]!r}") from self._errors[0]

    def serve_until_done(self) -> None:
        """Body of a dedicated server rank: wait until every worker has sent ``end``."""
        self.join_servers()

    def run_epochs(self, dataset, epochs: int = 1, verbose: int = 0, max_steps: int | None = None, fault_hook=None,
                   checkpoint_dir: str | None = None, checkpoint_every: int = 0, resume: bool = False,
                   phase_times: bool | None = None):
        """Workers do not wait for each other between steps: only start and end are collective. A checkpoint is
        written at the end (the state of an asynchronous run in flight is not a single point in time)."""
        if resume:
            self.try_resume(checkpoint_dir)
            self.on_weights_loaded()
        it_count, last = 0, None
        if self.is_worker:
            n_local = dataset.num_batches()
            steps = n_local if max_steps is None else min(n_local, max_steps)
            self.sync_device()
            self.comm.barrier()
            self._beat("first step")
            t0 = time.time()
            for ep in range(epochs):
                it = dataset.iterate(self.comm.device)
                for _ in range(steps):
                    x, y = next(it)
                    if fault_hook is not None:
                        fault_hook(self.comm.rank, it_count)
                    ts = time.time()
                    last = self.train_step(x, y)
                    if verbose and it_count % 50 == 0:
                        print(it_count, flush=True)
                    if self.metrics.enabled:
                        self.metrics.log(step=it_count, epoch=ep, loss=float(last), step_s=time.time() - ts,
                                         examples=int(y.shape[0]), version=max(self._versions_seen.values()))
                    it_count += 1
                    self._beat(f"step {it_count}")
        else:
            self.comm.barrier()
            t0 = time.time()
        if self.watchdog is not None:
            self.watchdog.pause()        # a finished worker waits for the slowest one below
        self.finalize()
        self.sync_device()
        self.comm.barrier()
        res = {"elapsed": time.time() - t0, "steps": it_count, "loss": float(last) if last is not None else None,
               "mean_staleness": (sum(self.staleness_log) / max(1, len(self.staleness_log))),
               "max_staleness": max(self.staleness_log, default=0)}
        if checkpoint_dir:
            self.save_checkpoint(checkpoint_dir, it_count)
        return res
