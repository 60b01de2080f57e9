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

import os
import time

import torch
import torch.distributed as dist

from ..comm.backend import Comm
from ..engine.optimizer import FlatOptimizer
from ..engine.replica import Replica, TrainConfig
from ..obs.metrics import MetricsLogger
from ..obs.trace import PhaseTimer


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
        self.watchdog = None            # comm.watchdog.Watchdog, set by the launcher
        self.phase_ms: dict[str, float] = {}   # per-phase device time of the last run_epochs (ms per step)
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

    def export_master(self) -> torch.Tensor:
        """The whole fp32 master arena (a collective call where the strategy shards it)."""
        return self.flat.master.detach().clone()

    def import_master(self, full: torch.Tensor) -> None:
        self.flat.write_master(full)

    def on_weights_loaded(self) -> None:
        """Called after ``flat.master`` has been rewritten from outside (weight import, checkpoint resume): a
        strategy that keeps parameter state apart from the replica's arena copies it from there."""

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

    def _beat(self, what: str) -> None:
        if self.watchdog is not None:
            self.watchdog.beat(what)

    def save_checkpoint(self, directory: str, step: int) -> None:
        from ..utils.checkpoint import save_checkpoint
        if self.watchdog is not None:
            self.watchdog.pause()
        save_checkpoint(directory, self, step=step)
        if self.watchdog is not None:
            self.watchdog.resume()

    def try_resume(self, directory: str | None) -> int:
        """Loads the checkpoint in ``directory`` if there is one; returns the number of steps it had completed."""
        if not directory or not os.path.exists(os.path.join(directory, "model.pt")):
            return 0
        from ..utils.checkpoint import load_checkpoint
        return load_checkpoint(directory, self)

    def run_epochs(self, dataset, epochs: int = 1, verbose: int = 0, max_steps: int | None = None,
                   fault_hook=None, checkpoint_dir: str | None = None, checkpoint_every: int = 0,
                   resume: bool = False, phase_times: bool | None = None) -> dict:
        """``checkpoint_dir`` + ``checkpoint_every=N``: a checkpoint after every N completed steps and at the end;
        ``resume``: continue from the checkpoint in ``checkpoint_dir`` -- the batches of the steps it had completed
        are skipped, so the run continues with the batch the interrupted one would have seen next.
        ``phase_times`` (default: when a metrics file is open and the step is not graph-captured): bracket the
        phases of every step with device events; their per-step averages land in ``self.phase_ms``, the result
        and the metrics log."""
        dev = self.comm.device
        steps_per_epoch = self.agree_on_steps(dataset.num_batches())
        if max_steps is not None:
            steps_per_epoch = min(steps_per_epoch, max_steps)
        start = self.try_resume(checkpoint_dir) if resume else 0
        it_count = 0
        last_loss = None
        step = self.step_callable()
        if phase_times is None:
            phase_times = self.metrics.enabled
        timer = PhaseTimer(dev) if (phase_times and not (self.cfg.graph and dev.type == "cuda")) else None
        self.sync_device()
        self.comm.barrier()
        self._beat("first step")
        t0 = time.time()
        if timer is not None:
            timer.__enter__()
        try:
            for ep in range(epochs):
                if (ep + 1) * steps_per_epoch <= start:      # an epoch the checkpoint had completed
                    it_count += steps_per_epoch
                    continue
                if hasattr(dataset, "_epoch"):
                    dataset._epoch = ep                      # the shuffle order is a function of (seed, epoch)
                it = dataset.iterate(dev)
                for i in range(steps_per_epoch):
                    batch = next(it)
                    if it_count < start:
                        it_count += 1
                        continue
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
                    self._beat(f"step {it_count}")
                    if self.cfg.verify_replicas and it_count % self.cfg.verify_replicas == 0:
                        self.check_replicas(it_count)
                    if checkpoint_dir and checkpoint_every and it_count % checkpoint_every == 0:
                        self.save_checkpoint(checkpoint_dir, it_count)
                    if timer is not None and it_count % 50 == 0:
                        self.metrics.log(event="phases", step=it_count, phase_ms=timer.report())
        finally:
            if timer is not None:
                timer.__exit__(None, None, None)
        self.finalize()
        self.sync_device()
        self.comm.barrier()
        elapsed = time.time() - t0
        res = {"elapsed": elapsed, "steps": it_count, "loss": float(last_loss) if last_loss is not None else None,
               "resumed_from": start}
        if timer is not None:
            self.phase_ms = timer.report()
            res["phase_ms"] = self.phase_ms
            if self.phase_ms and it_count > start:
                step_ms = elapsed * 1e3 / (it_count - start)
                res["exposed_comm_fraction"] = sum(v for k, v in self.phase_ms.items()
                                                   if k in ("grad_sync_wait", "push_wait", "pull", "push_pull", "gossip")) / step_ms
        if checkpoint_dir:
            self.save_checkpoint(checkpoint_dir, it_count)
        return res

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
