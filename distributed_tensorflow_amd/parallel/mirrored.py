"""Decentralized strategies: ``keras`` (mirrored), ``custom`` and ``graph``.

``keras``  -- synchronous mirrored data parallelism: every replica computes
gradients on its own batch, gradient buckets are all-reduced on RCCL while
backward is still running, and every replica applies the same fused optimizer
step. It stands in for ``MultiWorkerMirroredStrategy(RING)`` + ``model.fit``
of the reference (reference decentralized/native/dist_keras.py:27-32,42,77-78).

``custom`` -- the same training loop, but each bucket is reduced by one of the
hand-written algorithms of ``comm/algorithms.py`` (direct one-hop
reduce-scatter/all-gather, ring, multi-ring, tree, one-shot) on a side stream.
The reference only reserves the name (reference initializer.py:179-181).

``graph``  -- decentralized SGD on a sparse communication graph: each replica
takes a local optimizer step, then averages its parameters with ``degree``
neighbours of a circulant graph (degree <= 7 maps one-to-one onto the xGMI
links of a GPU). The reference reserves the name and a commented-out
``-d degree`` flag (reference initializer.py:90-92,175-177).
"""
from __future__ import annotations


import torch

from ..comm import algorithms as algos
from ..comm.backend import Comm
from ..engine.optimizer import FlatOptimizer
from ..engine.replica import TrainConfig
from ..ops import bucket as bops
from ..obs.trace import phase
from .base import Strategy


class MirroredStrategy(Strategy):
    name = "keras"
    replicas_agree = True
    graph_capturable = True

    def __init__(self, model, comm: Comm, cfg: TrainConfig | None = None, optimizer=None, metrics=None):
        super().__init__(model, comm, cfg, optimizer, metrics)
        self.optimizer = FlatOptimizer(self.flat, optimizer)
        self.broadcast_initial(0)
        # 'mean' loss: average of per-replica means; 'sum' loss: sum over the global batch
        self._scale = 1.0 / comm.world if self.cfg.loss_reduction == "mean" else 1.0

    def _on_ready(self, b):
        return self.comm.all_reduce(self.flat.grad[b.start:b.end], async_op=True)

    def train_step(self, x, y):
        loss = self.replica.forward_backward(x, y, self._on_ready if self.comm.world > 1 else None, self._scale)
        with phase("grad_sync_wait"):
            for w in self.replica.finish_backward():
                if w is not None:
                    w.wait()
        with phase("optimizer"):
            self.optimizer.step()
            self.apply_sparse()
        return loss


class CustomAllReduceStrategy(Strategy):
    name = "custom"
    replicas_agree = True

    def __init__(self, model, comm: Comm, cfg: TrainConfig | None = None, optimizer=None, metrics=None,
                 algorithm: str = "auto"):
        cfg = cfg or TrainConfig()
        cfg.shard_ways = max(cfg.shard_ways, comm.world)  # buckets must split into `world` aligned shards
        super().__init__(model, comm, cfg, optimizer, metrics)
        self.algorithm = algorithm
        self.optimizer = FlatOptimizer(self.flat, optimizer)
        self.broadcast_initial(0)
        self._scale = 1.0 / comm.world if self.cfg.loss_reduction == "mean" else 1.0
        biggest = max(b.numel for b in self.planner.buckets)
        self.used_algorithms: list[str] = []
        # 'ipc' / 'ipc-<schedule>': kernels on HIP IPC mapped peer arenas instead of grouped send/recv
        self._peers = None
        self._scratch = None
        if algorithm.startswith("ipc") and comm.world > 1:
            if comm.device.type != "cuda":
                raise ValueError("the peer-pointer algorithms need GPU devices")
            from ..comm.ipc import PeerGroup
            self._peers = PeerGroup(comm)
            self._peers.register(self.flat.grad)
            self._ipc_schedule = algorithm.partition("-")[2] or "auto"
        elif comm.world > 1:
            self._scratch = torch.empty(biggest, dtype=self.flat.grad.dtype, device=comm.device)

    def _reduce(self, view):
        if self._peers is not None:
            return "ipc-" + self._peers.all_reduce(view, self._ipc_schedule, base=self.flat.grad)
        return algos.all_reduce(self.comm, view, self.algorithm, self._scratch)

    def _on_ready(self, b):
        view = self.flat.grad[b.start:b.end]
        side = self.side_stream()
        if side is None:
            self.used_algorithms.append(self._reduce(view))
            return None
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            self.used_algorithms.append(self._reduce(view))
            done = torch.cuda.Event()
            done.record(side)
        return done

    def train_step(self, x, y):
        self.used_algorithms.clear()
        loss = self.replica.forward_backward(x, y, self._on_ready if self.comm.world > 1 else None, self._scale)
        with phase("grad_sync_wait"):
            for ev in self.replica.finish_backward():
                if ev is not None:
                    torch.cuda.current_stream(self.comm.device).wait_event(ev)
        with phase("optimizer"):
            self.optimizer.step()
        if self._peers is not None:
            self._peers.check_async()     # a barrier that timed out means partial sums: stop, do not train on
        return loss

    def save_checkpoint(self, directory: str, step: int) -> None:
        """A barrier that timed out leaves partial sums in the gradients already applied: look at the status word
        itself (a host wait, once per checkpoint) before anything is written, so that no checkpoint can hold them."""
        if self._peers is not None:
            self._peers.check()
        super().save_checkpoint(directory, step)

    def finalize(self) -> None:
        if self._peers is not None:
            self._peers.check()
            self._peers.close()
            self._peers = None


def circulant_neighbours(rank: int, world: int, degree: int) -> list[int]:
    """Neighbours of ``rank`` in the undirected circulant graph of ``world`` nodes in which every node has
    exactly ``degree`` neighbours: offsets +-1 ... +-(degree // 2), plus the antipodal node ``rank + world / 2``
    when ``degree`` is odd. ``j`` is in the list of ``i`` exactly when ``i`` is in the list of ``j``, for every
    degree -- a one-directional edge would leave a send without its matching receive. A degree of
    ``world - 1`` or more gives the complete graph; an odd degree on an odd number of nodes does not exist
    (the degrees of a graph sum to an even number) and raises."""
    degree = max(0, min(int(degree), world - 1))
    if degree == world - 1:
        return [(rank + k) % world for k in range(1, world)]
    if degree % 2 == 1 and world % 2 == 1:
        raise ValueError(f"no {degree}-regular graph on {world} nodes: an odd degree needs an even number of nodes")
    out: list[int] = []
    for k in range(1, degree // 2 + 1):
        out += [(rank + k) % world, (rank - k) % world]
    if degree % 2 == 1:
        out.append((rank + world // 2) % world)
    assert len(set(out)) == degree and rank not in out
    return out


def graph_is_undirected(world: int, degree: int) -> bool:
    """Every edge of the circulant graph is seen from both of its ends."""
    nb = [set(circulant_neighbours(r, world, degree)) for r in range(world)]
    return all(i in nb[j] for i in range(world) for j in nb[i])


class GraphStrategy(Strategy):
    name = "graph"

    def __init__(self, model, comm: Comm, cfg: TrainConfig | None = None, optimizer=None, metrics=None,
                 degree: int = 2, gossip_every: int = 1):
        super().__init__(model, comm, cfg, optimizer, metrics)
        self.optimizer = FlatOptimizer(self.flat, optimizer)
        self.broadcast_initial(0)
        self.degree = max(0, min(int(degree), comm.world - 1))
        self.neighbours = circulant_neighbours(comm.rank, comm.world, self.degree)   # raises for odd x odd
        if not graph_is_undirected(comm.world, self.degree):
            raise AssertionError(f"graph of degree {self.degree} on {comm.world} nodes has a one-directional edge")
        self.gossip_every = max(1, gossip_every)
        self._recv = [torch.empty_like(self.flat.master) for _ in self.neighbours]
        self._steps = 0

    def gossip(self) -> None:
        """x_i <- (x_i + sum_{j in N(i)} x_j) / (deg + 1): a doubly stochastic mixing step."""
        if not self.neighbours:
            return
        m = self.flat.master
        works = self.comm.exchange([(m, n) for n in self.neighbours],
                                   [(r, n) for r, n in zip(self._recv, self.neighbours)])
        for w in works:
            w.wait()
        w_ = 1.0 / (len(self.neighbours) + 1)
        bops.nary_axpy(m, self._recv, [w_] * len(self._recv), w_dst=w_)
        self.flat.refresh_shadow()

    def train_step(self, x, y):
        loss = self.replica.forward_backward(x, y, None, 1.0)
        self.replica.finish_backward()
        with phase("optimizer"):
            self.optimizer.step()
        self._steps += 1
        if self._steps % self.gossip_every == 0:
            with phase("gossip"):
                self.gossip()
        return loss

    def finalize(self) -> None:
        """Consensus: average the replicas so that every node ends with the same model."""
        if self.comm.world > 1:
            self.comm.all_reduce(self.flat.master)
            self.flat.master.mul_(1.0 / self.comm.world)
            self.flat.refresh_shadow()

    def disagreement(self) -> float:
        """max over ranks of ||x_i - mean(x)|| / ||mean(x)|| (0 for agreeing replicas)."""
        mean = self.flat.master.clone()
        self.comm.all_reduce(mean)
        mean.mul_(1.0 / self.comm.world)
        d = float((self.flat.master - mean).norm() / mean.norm().clamp_min(1e-12))
        return self.comm.all_reduce_scalar(d, op=torch.distributed.ReduceOp.MAX) if self.comm.world > 1 else d


# ------------------------------------------------------------------------- keras facade
def _strategy_for(model, strategy):
    if strategy is not None:
        return strategy
    st = getattr(model, "_dtf_strategy", None)
    if st is None:
        dev = next(model.parameters()).device
        comp = getattr(model, "_compiled", None) or {}
        st = MirroredStrategy(model, Comm(0, 1, dev), optimizer=comp.get("optimizer"))
        model._dtf_strategy = st
    return st


def fit(model, dataset, epochs: int = 1, verbose: int = 0, strategy=None, max_steps: int | None = None) -> dict:
    """``model.fit``: trains ``epochs`` passes and returns ``{'elapsed', 'steps', 'loss'}``."""
    st = _strategy_for(model, strategy)
    return st.run_epochs(dataset, epochs=epochs, verbose=verbose, max_steps=max_steps)


def evaluate(model, dataset, strategy=None) -> list[float]:
    """``model.evaluate``: returns ``[loss, accuracy]`` like Keras with ``metrics=['accuracy']``."""
    st = _strategy_for(model, strategy)
    loss, acc = st.evaluate(dataset)
    return [loss, acc]
