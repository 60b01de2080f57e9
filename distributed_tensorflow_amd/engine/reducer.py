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

import ctypes
import time
from typing import Any, Callable

import torch

from ..obs.trace import phase
from ..ops import _lib
from ..ops import bucket as bops
from .flat import Bucket, BucketPlanner, FlatParams


def _side_grads():
    from ..ops.conv import side_grads
    return side_grads


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
        self._plans: dict = {}
        self.hook_host_s = 0.0             # host time spent inside on_ready callbacks (reset by the caller)
        self.hook_calls = 0

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
        _side_grads().open()               # weight gradients may run on a side stream until finish()

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
                # its own phase inside ``backward``: what a collective costs where it is STARTED (a staged or
                # synchronous transfer blocks here) is communication, not backward compute
                t0 = time.perf_counter()
                with phase("comm_in_hook"):
                    self.works[b.index] = self.on_ready(b)
                self.hook_host_s += time.perf_counter() - t0
                self.hook_calls += 1
            self._next += 1

    # ------------------------------------------------------------------ pack
    def _plan(self, b: Bucket, dtype: torch.dtype):
        """The step-invariant half of a bucket's pack call (offset and length tables, dtype codes), built once per
        (bucket, gradient dtype); only the pointer table changes from step to step."""
        key = (b.index, dtype)
        plan = self._plans.get(key)
        if plan is None:
            n = len(b.slots)
            plan = (n, (ctypes.c_longlong * n)(*[s.offset for s in b.slots]),
                    (ctypes.c_longlong * n)(*[s.numel for s in b.slots]), (ctypes.c_void_p * n)(),
                    _lib.dtype_code(dtype), _lib.dtype_code(self.flat.grad.dtype),
                    [tuple(s.stride) for s in b.slots])
            self._plans[key] = plan
        return plan

    @torch.no_grad()
    def _pack(self, b: Bucket) -> None:
        flat = self.flat
        grads = [s.param.grad for s in b.slots]
        side = _side_grads()
        for g in grads:                    # gradients computed on the side stream: wait for their kernels
            if g is not None:
                side.wait(g)
        g0 = grads[0]
        if flat.grad.is_cuda and g0 is not None and all(g is not None and g.dtype == g0.dtype for g in grads):
            # fast path (every step of an ordinary run): all gradients present, one dtype
            n, offs, nums, ptrs, src_code, dst_code, strides = self._plan(b, g0.dtype)
            for k, g in enumerate(grads):
                if g.stride() != strides[k]:
                    g = self._like_param(g, b.slots[k])
                    grads[k] = g
                ptrs[k] = g.data_ptr()
            rc = _lib.kernels().dtf_bucket_flatten(n, ptrs, offs, nums, _lib.ptr(flat.grad), src_code, dst_code, None,
                                                   ctypes.c_float(self.grad_scale), _lib.stream_ptr(flat.device))
            _lib.check(rc, "bucket_flatten")
            for s in b.slots:
                s.param.grad = None
            return
        groups: dict[torch.dtype, tuple[list, list]] = {}
        for s, g in zip(b.slots, grads):
            if g is None:            # parameter took no part in this step
                flat.grad[s.offset:s.offset + s.numel].zero_()
                continue
            g = self._like_param(g, s)
            ts, offs = groups.setdefault(g.dtype, ([], []))
            ts.append(g.as_strided((g.numel(),), (1,)))
            offs.append(s.offset)
            s.param.grad = None
        for ts, offs in groups.values():
            # every gradient is a dense block laid out like its parameter: pack its raw memory order
            bops.flatten(ts, flat.grad, offs, scale=self.grad_scale)

    @staticmethod
    def _like_param(g: torch.Tensor, s) -> torch.Tensor:
        """``g`` in the memory order of its parameter (strides of size-1 dimensions do not matter)."""
        if any(gs != ss for gs, ss, n in zip(g.stride(), s.stride, s.shape) if n != 1):
            return torch.empty_strided(tuple(s.shape), s.stride, dtype=g.dtype, device=g.device).copy_(g)
        return g

    # ---------------------------------------------------------------- finish
    def finish(self) -> list[Any]:
        """Packs and launches every bucket that backward did not complete (unused parameters),
        then returns the per-bucket work handles in bucket order."""
        if not self._in_step:
            return []
        for i in range(len(self._ready)):
            self._ready[i] = True
        self._drain()
        _side_grads().close()
        self._in_step = False
        return self.works
