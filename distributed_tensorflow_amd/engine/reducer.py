"""Backward-overlapped gradient bucketing.

Per-parameter autograd hooks count gradients into buckets; the moment a
bucket's last gradient exists, ONE multi-tensor kernel packs (and scales and
casts) the bucket into its slice of the flat gradient arena and the strategy's
``on_ready(bucket)`` callback starts that bucket's collective while backward
keeps running. Buckets are launched strictly in index order on every rank, so
collectives are issued in the same order everywhere.

The reference has no overlap at all: a worker computes all gradients, pickles
them, sends them, and blocks for the reply (reference
centralized/client.py:83-90).
"""
from __future__ import annotations

from typing import Any, Callable

import torch

from ..ops import bucket as bops
from .flat import Bucket, BucketPlanner, FlatParams


class GradReducer:
    def __init__(self, flat: FlatParams, planner: BucketPlanner,
                 on_ready: Callable[[Bucket], Any] | None = None, grad_scale: float = 1.0):
        self.flat, self.planner = flat, planner
        self.on_ready = on_ready
        self.grad_scale = grad_scale
        self._handles = []
        self._pending: list[int] = []
        self._ready: list[bool] = []
        self._next = 0
        self.works: list[Any] = []
        self.enabled = True
        self._in_step = False

    # ----------------------------------------------------------------- hooks
    def attach(self) -> None:
        for s in self.flat.slots:
            self._handles.append(s.param.register_post_accumulate_grad_hook(self._hook))

    def detach(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles.clear()

    def begin_step(self) -> None:
        self._pending = [len(b.slots) for b in self.planner.buckets]
        self._ready = [False] * len(self.planner.buckets)
        self._seen = set()
        self._next = 0
        self.works = [None] * len(self.planner.buckets)
        self._in_step = True

    def _hook(self, param: torch.Tensor) -> None:
        if not (self.enabled and self._in_step):
            return
        pid = id(param)
        if pid in self._seen:
            return
        self._seen.add(pid)
        b = self.planner.slot_bucket[pid]
        self._pending[b.index] -= 1
        if self._pending[b.index] == 0:
            self._ready[b.index] = True
            self._drain()

    def _drain(self) -> None:
        nb = len(self.planner.buckets)
        while self._next < nb and self._ready[self._next]:
            b = self.planner.buckets[self._next]
            self._pack(b)
            if self.on_ready is not None:
                self.works[b.index] = self.on_ready(b)
            self._next += 1

    # ------------------------------------------------------------------ pack
    @torch.no_grad()
    def _pack(self, b: Bucket) -> None:
        flat = self.flat
        groups: dict[torch.dtype, tuple[list, list]] = {}
        missing = []
        for s in b.slots:
            g = s.param.grad
            if g is None:
                missing.append(s)
                continue
            if any(gs != ss for gs, ss, n in zip(g.stride(), s.stride, s.shape) if n != 1):
                g = torch.empty_strided(tuple(s.shape), s.stride, dtype=g.dtype, device=g.device).copy_(g)
            ts, offs = groups.setdefault(g.dtype, ([], []))
            ts.append(g)
            offs.append(s.offset)
            s.param.grad = None
        for s in missing:  # parameter took no part in this step
            flat.grad[s.offset:s.offset + s.numel].zero_()
        for dt, (ts, offs) in groups.items():
            # every gradient is a dense block laid out like its parameter: pack its raw memory order
            bops.flatten([t.as_strided((t.numel(),), (1,)) for t in ts], flat.grad, offs, scale=self.grad_scale)

    # ---------------------------------------------------------------- finish
    def finish(self) -> list[Any]:
        """Packs and launches every bucket that backward did not complete (unused parameters),
        then returns the per-bucket work handles in bucket order."""
        if not self._in_step:
            return []
        for i in range(len(self._ready)):
            self._ready[i] = True
        self._drain()
        self._in_step = False
        return self.works
