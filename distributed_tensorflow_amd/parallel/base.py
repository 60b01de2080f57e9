"""Common machinery of the data-parallel strategies.

A strategy owns a :class:`Replica` (model + flat arenas), a :class:`Comm` and a
:class:`FlatOptimizer`, and implements ``train_step(x, y)``. ``run_epochs``
is the shared outer loop: it equalises the number of steps across ranks
(shards of a dataset can differ by one batch, which in the reference deadlocks
the synchronous barrier -- reference centralized/server.py:95,113 with the
sharding of initializer.py:44), reports progress every 50 iterations like the
reference worker (centralized/client.py:92-93), and returns the wall-clock time
between the first step and the last, the quantity the reference prints as
``elapsed time`` (server.py:116-119, decentralized/native/dist_keras.py:41-51).
"""
from __future__ import annotations

import time

import torch
import torch.distributed as dist

from ..comm.backend import Comm
from ..engine.optimizer import FlatOptimizer
from ..engine.replica import Replica, TrainConfig
from ..obs.metrics import MetricsLogger


class ReplicaDivergence(RuntimeError):
    """The replicas of a strategy that keeps them bit-identical no longer agree."""


class Strategy:
    name = "base"
    replicas_agree = False      # every rank holds the same master weights after each step
    graph_capturable = False    # train_step is stream-ordered end to end (no host round trip inside)

    def __init__(self, model, comm: Comm, cfg: TrainConfig | None = None, optimizer=None,
                 metrics: MetricsLogger | None = None):
        self.comm = comm
        self.cfg = cfg or TrainConfig()
        torch.manual_seed(self.cfg.seed)
        self.replica = Replica(model, comm.device, self.cfg)
        self.flat = self.replica.flat
        self.planner = self.replica.planner
        self.optimizer_spec = optimizer
        self.metrics = metrics or MetricsLogger(None, rank=comm.rank)
        self.optimizer: FlatOptimizer | None = None
        self._side_stream = None
        # row-sharded embedding tables are parameter-server state outside the dense arenas
        from .sharded_embedding import ShardedEmbedding
        self.sparse_modules = [m for m in self.replica.model.modules() if isinstance(m, ShardedEmbedding)]

    # ------------------------------------------------------------ lifecycle
    def broadcast_initial(self, src: int = 0) -> None:
        """Every replica starts from rank ``src``'s weights (the parameter server's initial
        weight message, reference centralized/server.py:81-84, and the variable broadcast of
        the mirrored strategy)."""
        if self.comm.world > 1:
            self.comm.broadcast(self.flat.master, src)
            self.flat.refresh_shadow()
            for b in self.replica.model.buffers():
                if b.is_floating_point():
                    self.comm.broadcast(b, src)

    def train_step(self, x, y) -> torch.Tensor:  # pragma: no cover - interface
        raise NotImplementedError

    def step_callable(self):
        """``train_step``, or its hipGraph-captured form when ``cfg.graph`` asks for it."""
        if self.cfg.graph and self.comm.device.type == "cuda":
            if not self.graph_capturable:
                raise ValueError(f"strategy {self.name!r} has host-side hand-offs inside a step and cannot be "
                                 "captured into a graph")
            if getattr(self, "_graphed", None) is None:
                from ..engine.graph import GraphedStep
                self._graphed = GraphedStep(self.train_step, self.comm.device)
            return self._graphed
        return self.train_step

    def apply_sparse(self) -> None:
        """Parameter-server apply of the sparse (embedding) gradients pushed during backward."""
        if self.sparse_modules:
            scale = 1.0 / self.comm.world if self.cfg.loss_reduction == "mean" else 1.0
            for m in self.sparse_modules:
                m.apply(scale)

    def check_replicas(self, step: int = -1) -> None:
        """Debug check of the synchronous strategies: the fp64 (sum, sum of squares) of the master
        weights must be identical on every rank -- the all-reduce / parameter-server apply is
        deterministic and identical across replicas, so any difference is a lost or doubled
        update. Costs one 4-number all-reduce; raises :class:`ReplicaDivergence` on every rank."""
        if not self.replicas_agree or self.comm.world == 1:
            return
        c = self.flat.checksum()                       # fp64, on the arena's device
        t = torch.cat([c, -c])
        self.comm.all_reduce(t, op=dist.ReduceOp.MAX)  # max and (negated) min in one collective
        t, c = t.cpu(), c.cpu()
        hi, lo = t[:2], -t[2:]
        if not torch.equal(hi, lo):
            raise ReplicaDivergence(
                f"step {step}: master weights differ across ranks (checksum max {hi.tolist()} != min {lo.tolist()}; "
                f"this rank {c.tolist()})")

    def finalize(self) -> None:
        """Brings every replica to the final model (no-op for strategies whose replicas agree)."""

    def sync_device(self) -> None:
        if self.comm.device.type == "cuda":
            torch.cuda.synchronize(self.comm.device)

    # ------------------------------------------------------------ outer loop
    def agree_on_steps(self, n_local: int) -> int:
        """min over ranks of the local number of batches."""
        if self.comm.world == 1:
            return n_local
        return int(self.comm.all_reduce_scalar(float(n_local), op=dist.ReduceOp.MIN))

    def run_epochs(self, dataset, epochs: int = 1, verbose: int = 0, max_steps: int | None = None,
                   fault_hook=None) -> dict:
        dev = self.comm.device
        steps_per_epoch = self.agree_on_steps(dataset.num_batches())
        if max_steps is not None:
            steps_per_epoch = min(steps_per_epoch, max_steps)
        it_count = 0
        last_loss = None
        step = self.step_callable()
        self.sync_device()
        self.comm.barrier()
        t0 = time.time()
        for ep in range(epochs):
            it = dataset.iterate(dev)
            for i in range(steps_per_epoch):
                batch = next(it)
                x, y = (batch[0] if len(batch) == 2 else tuple(batch[:-1])), batch[-1]
                if fault_hook is not None:
                    fault_hook(self.comm.rank, it_count)
                ts = time.time()
                last_loss = step(x, y)
                if verbose and it_count % 50 == 0 and self.comm.rank == 0:
                    print(it_count, flush=True)
                if self.metrics.enabled:
                    self.metrics.log(step=it_count, epoch=ep, loss=float(last_loss), step_s=time.time() - ts,
                                     examples=int(y.shape[0]))
                it_count += 1
                if self.cfg.verify_replicas and it_count % self.cfg.verify_replicas == 0:
                    self.check_replicas(it_count)
        self.finalize()
        self.sync_device()
        self.comm.barrier()
        elapsed = time.time() - t0
        return {"elapsed": elapsed, "steps": it_count, "loss": float(last_loss) if last_loss is not None else None}

    # ------------------------------------------------------------ helpers
    def side_stream(self):
        if self._side_stream is None and self.comm.device.type == "cuda":
            self._side_stream = torch.cuda.Stream(self.comm.device)
        return self._side_stream

    def evaluate(self, dataset, reduce: bool = True, max_batches: int | None = None) -> tuple[float, float]:
        loss, acc, n = self.replica.evaluate(dataset, max_batches)
        if reduce and self.comm.world > 1:
            t = torch.tensor([loss * n, acc * n, float(n)], dtype=torch.float64 if self.comm.device.type == "cpu"
                             else torch.float32, device=self.comm.device)
            self.comm.all_reduce(t)
            loss, acc = float(t[0] / t[2]), float(t[1] / t[2])
        return loss, acc
