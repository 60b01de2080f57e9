"""Row-sharded embedding tables on the parameter-server shards.

The Wide&Deep configuration keeps ~200 GB of embedding rows on the parameter server. Here the
server state is the union of the GPUs' HBM: rank r owns rows ``[r*R, (r+1)*R)`` of every table
(``R = ceil(V / W)``), in fp32, together with the optimizer state of those rows. A training
step is the parameter-server cycle, restricted to the rows a batch touches:

  pull   ids are bucketed by owner and exchanged (all-to-all); every owner gathers the
         requested rows from its shard (one kernel) and sends them back (all-to-all);
  push   row gradients travel the same way in reverse;
  apply  the owner sums the gradients per unique id (sort + segment sum, reproducible) and
         updates only those rows with a sparse optimizer kernel (SGD / Adagrad / lazy Adam).

So a step moves O(batch) rows over xGMI and never touches the other ~200 GB. The reference
parameter server has no sparse path: it ships every weight of the model to every worker on
every step (reference centralized/server.py:96,101,104-105).
"""
from __future__ import annotations

import math

import torch
import torch.distributed as dist
import torch.nn as nn

from ..comm.backend import Comm
from ..ops import embedding as E


def _all_to_all_v(comm: Comm, send: torch.Tensor, send_counts: list[int], recv_counts: list[int]) -> torch.Tensor:
    """Variable-size all-to-all of rows (dim 0)."""
    out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if comm.world == 1:
        out.copy_(send)
        return out
    if not getattr(comm, "composed", comm.device.type != "cuda"):
        dist.all_to_all_single(out, send.contiguous(), recv_counts, send_counts, group=comm.group)
        return out
    so = ro = 0
    sends, recvs = [], []
    for r in range(comm.world):
        s, c = send[so:so + send_counts[r]], out[ro:ro + recv_counts[r]]
        so += send_counts[r]
        ro += recv_counts[r]
        if r == comm.rank:
            c.copy_(s)
            continue
        if s.numel():
            sends.append((s.contiguous(), r))
        if c.numel():
            recvs.append((c, r))
    for w in comm.exchange(sends, recvs):
        w.wait()
    return out


class _Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, module, ids, plan=None):
        rows, plan = module._pull(ids, plan)
        ctx.module, ctx.plan = module, plan
        return rows

    @staticmethod
    def backward(ctx, g):
        ctx.module._push(g, ctx.plan)
        return torch.zeros((), device=g.device), None, None, None


class ShardedEmbedding(nn.Module):
    """``[V, D]`` fp32 table row-sharded over ``comm``; parameters are NOT registered with the
    dense flat arenas -- the table is parameter-server state updated by :meth:`apply`."""

    def __init__(self, num_rows: int, dim: int, comm: Comm, optimizer: str = "adagrad", lr: float = 0.05,
                 init_std: float = 0.01, eps: float = 1e-7, seed: int = 0, init: str = "normal"):
        super().__init__()
        if dim % 4:
            raise ValueError("embedding dim must be a multiple of 4")
        self.comm, self.V, self.D = comm, num_rows, dim
        self.rows_per_shard = math.ceil(num_rows / comm.world)
        self.row_base = comm.rank * self.rows_per_shard
        n_local = max(0, min(self.rows_per_shard, num_rows - self.row_base))
        dev = comm.device
        g = torch.Generator(device="cpu").manual_seed(seed + 7919 * comm.rank)
        if init == "zeros" or n_local * dim > (1 << 31):
            # very large shards: fill on the device in slabs (a host-side normal_() of tens of GB is slow)
            self.table = torch.empty(n_local, dim, dtype=torch.float32, device=dev)
            if init == "zeros":
                self.table.zero_()
            else:
                gd = torch.Generator(device=dev).manual_seed(seed + 7919 * comm.rank) if dev.type == "cuda" else g
                step = max(1, (1 << 28) // dim)
                for a in range(0, n_local, step):
                    self.table[a:a + step].normal_(0, init_std, generator=gd)
        else:
            self.table = (torch.randn(n_local, dim, generator=g) * init_std).to(dev)
        self.kind = {"sgd": E.SGD, "adagrad": E.ADAGRAD, "adam": E.ADAM}[optimizer]
        self.m = torch.zeros_like(self.table) if self.kind == E.ADAM else None
        self.v = torch.zeros_like(self.table) if self.kind in (E.ADAGRAD, E.ADAM) else None
        self.lr, self.eps, self.beta1, self.beta2 = lr, eps, 0.9, 0.999
        self.step_count = 0
        self._anchor = nn.Parameter(torch.zeros(()), requires_grad=True)  # makes the lookup differentiable
        self._anchor.dtf_exclude = True
        self._pending: list[tuple[torch.Tensor, torch.Tensor]] = []
        self.rows_pulled = 0

    # ------------------------------------------------------------------ pull
    def plan(self, ids: torch.Tensor):
        """Routing of one batch of ids: who owns which, in what order they travel, and the ids every owner has to
        serve. It depends on the ids and the shard size only, so tables of the same height that are looked up with
        the same ids (Wide&Deep's deep and wide tables) share ONE plan: one id exchange and ONE host read of the
        per-owner counts per step instead of one per table (the variable-size all-to-all needs its split sizes on the
        host; that read is the only host synchronisation of a lookup)."""
        comm, W = self.comm, self.comm.world
        flat = ids.reshape(-1).long()
        owner = torch.div(flat, self.rows_per_shard, rounding_mode="floor").clamp_(0, W - 1)
        order = torch.argsort(owner, stable=True)
        sorted_ids = flat[order]
        send_counts = torch.bincount(owner, minlength=W)
        if W > 1:
            recv_counts_t = torch.empty_like(send_counts)
            comm.all_to_all(recv_counts_t, send_counts)
            sc, rc = torch.stack([send_counts, recv_counts_t]).tolist()       # one device -> host read for both
        else:
            sc = rc = [flat.numel()]
        req = _all_to_all_v(comm, sorted_ids, sc, rc)                      # ids this shard must serve
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=order.device)
        return (order, sc, rc, req, inv, self.rows_per_shard, tuple(ids.shape))

    def _pull(self, ids: torch.Tensor, plan=None):
        if plan is None:
            plan = self.plan(ids)
        order, sc, rc, req, inv, rps, shape = plan
        if rps != self.rows_per_shard or shape != tuple(ids.shape):
            raise ValueError("lookup plan was made for another table height or batch shape")
        rows = E._gather(self.table, req, row_base=self.row_base)          # [m, D]
        got = _all_to_all_v(self.comm, rows, rc, sc)                       # rows for my ids, owner-sorted
        out = got.index_select(0, inv).view(*ids.shape, self.D)
        self.rows_pulled += int(ids.numel())
        return out, plan

    def forward(self, ids: torch.Tensor, plan=None) -> torch.Tensor:
        return _Lookup.apply(self._anchor, self, ids, plan)

    # ------------------------------------------------------------------ push
    def _push(self, g: torch.Tensor, plan) -> None:
        order, sc, rc, req = plan[:4]
        g2 = g.reshape(-1, self.D).float().index_select(0, order)          # owner-sorted gradients
        recv = _all_to_all_v(self.comm, g2, sc, rc)                        # gradients for the rows I own
        self._pending.append((req, recv))

    # ----------------------------------------------------------------- apply
    @torch.no_grad()
    def apply(self, grad_scale: float = 1.0) -> int:
        """Parameter-server apply step for this shard; returns the number of rows updated."""
        if not self._pending:
            return 0
        ids = torch.cat([p[0] for p in self._pending])
        grads = torch.cat([p[1] for p in self._pending])
        self._pending.clear()
        if ids.numel() == 0:
            return 0
        uniq, perm, seg = E.unique_segments(ids)
        gsum = E.segment_sum(grads, perm, seg)
        self.step_count += 1
        bc1 = 1.0 - self.beta1 ** self.step_count
        bc2 = 1.0 - self.beta2 ** self.step_count
        hyper = torch.tensor([self.lr, bc1, bc2, self.eps, self.beta1, self.beta2, grad_scale, 0.0],
                             dtype=torch.float32, device=self.table.device)
        E.sparse_apply(self.table, self.m, self.v, gsum, uniq, hyper, self.kind, row_base=self.row_base)
        return int(uniq.numel())

    def nbytes(self) -> int:
        n = self.table.numel() * 4
        for t in (self.m, self.v):
            if t is not None:
                n += t.numel() * 4
        return n

    # -------------------------------------------------------------- checkpoint
    def state_dict_shard(self) -> dict:
        sd = {"table": self.table.detach().cpu(), "row_base": self.row_base, "step": self.step_count}
        if self.m is not None:
            sd["m"] = self.m.detach().cpu()
        if self.v is not None:
            sd["v"] = self.v.detach().cpu()
        return sd

    def load_state_dict_shard(self, sd: dict) -> None:
        self.table.copy_(sd["table"].to(self.table.device))
        if self.m is not None and "m" in sd:
            self.m.copy_(sd["m"].to(self.table.device))
        if self.v is not None and "v" in sd:
            self.v.copy_(sd["v"].to(self.table.device))
        self.step_count = int(sd.get("step", 0))
