"""Flat parameter / gradient arenas.

All trainable parameters of a model live in ONE logical index space of ``N``
elements:

* ``master`` -- fp32 ``[N]``; fp32 parameters are views into it, and it is the
  optimizer's source of truth for every parameter;
* ``shadow`` -- bf16 ``[N]`` (only when the model has bf16 parameters); bf16
  parameters are views into it and the optimizer kernel refreshes it in the
  same pass that updates ``master``;
* ``grad``   -- ``[N]`` in ``grad_dtype`` (fp32 or bf16); gradient buckets are
  contiguous slices of it, so a collective runs on a bucket with no copy.

The layout is in REVERSE parameter order: backward produces the last layer's
gradients first, so bucket 0 (the lowest addresses) is the first to complete
and the first to be communicated.

The reference keeps weights as a Python list of per-variable arrays that is
pickled on every push and pull (reference centralized/server.py:96,101,104;
client.py:85-90); the arenas make that list a set of views and the wire format
a pointer + length.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, Sequence

import torch

ALIGN = 64  # elements; keeps every slot 128-byte (bf16) / 256-byte (fp32) aligned


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Slot:
    name: str
    param: torch.nn.Parameter
    offset: int
    numel: int
    shape: torch.Size
    dtype: torch.dtype
    stride: tuple = ()


def _dense_strides(p: torch.Tensor) -> tuple:
    """Strides to keep for ``p`` inside the arena: its own when it is a dense permutation of a
    contiguous block (e.g. channels_last convolution weights), contiguous ones otherwise."""
    st = tuple(p.stride())
    dims = sorted(range(p.dim()), key=lambda d: (st[d], p.shape[d]), reverse=True)
    expect = 1
    for d in reversed(dims):
        if p.shape[d] != 1 and st[d] != expect:
            return tuple(torch.empty(p.shape).stride())
        expect *= p.shape[d]
    return st


class FlatParams:
    """Re-homes the trainable parameters of ``module`` into flat arenas."""

    def __init__(self, named_params: Iterable[tuple[str, torch.nn.Parameter]], grad_dtype: torch.dtype = torch.float32,
                 shard_ways: int = 1):
        """``shard_ways``: every slot is padded to ``ALIGN * shard_ways`` elements so that any
        run of whole slots (a bucket) splits into ``shard_ways`` equal, aligned shards."""
        named = [(n, p) for n, p in named_params if p.requires_grad]
        if not named:
            raise ValueError("model has no trainable parameters")
        self.device = named[0][1].device
        for n, p in named:
            if p.device != self.device:
                raise ValueError(f"parameter {n} is on {p.device}, expected {self.device}")
            if p.dtype not in (torch.float32, torch.bfloat16):
                raise TypeError(f"parameter {n} has dtype {p.dtype}; arenas hold float32 and bfloat16")
        self.grad_dtype = grad_dtype
        self.shard_ways = max(1, int(shard_ways))
        self.align = ALIGN * self.shard_ways
        self.slots: list[Slot] = []
        off = 0
        for n, p in reversed(named):
            self.slots.append(Slot(n, p, off, p.numel(), p.shape, p.dtype, _dense_strides(p)))
            off += _round_up(p.numel(), self.align)
        self.used = off
        self.total = off
        self.has_bf16 = any(s.dtype == torch.bfloat16 for s in self.slots)
        self.master = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.shadow = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device) if self.has_bf16 else None
        self.grad = torch.zeros(self.total, dtype=grad_dtype, device=self.device)
        self.by_name = {s.name: s for s in self.slots}
        self.by_param = {id(s.param): s for s in self.slots}
        with torch.no_grad():
            for s in self.slots:
                mview = self._view(self.master, s)
                mview.copy_(s.param.detach())
                if s.dtype == torch.float32:
                    s.param.data = mview
                else:
                    sview = self._view(self.shadow, s)
                    sview.copy_(s.param.detach())
                    s.param.data = sview
                    s.param._dtf_master = mview     # api/keras_weights.py writes loaded weights here too

    @classmethod
    def from_module(cls, module: torch.nn.Module, **kw) -> "FlatParams":
        return cls(list(module.named_parameters()), **kw)

    # ------------------------------------------------------------------ views
    @staticmethod
    def _view(arena: torch.Tensor, s: Slot) -> torch.Tensor:
        return arena.as_strided(tuple(s.shape), s.stride, s.offset)

    def grad_view(self, s: Slot) -> torch.Tensor:
        return self._view(self.grad, s)

    def master_view(self, s: Slot) -> torch.Tensor:
        return self._view(self.master, s)

    def bind_grads_as_views(self) -> None:
        """Point every ``param.grad`` at its slice of the grad arena (autograd then
        accumulates in place). Requires ``grad_dtype`` equal to the param dtype."""
        for s in self.slots:
            if s.dtype != self.grad_dtype:
                raise TypeError("grad views need grad_dtype == parameter dtype for every parameter")
            s.param.grad = self.grad_view(s)

    def zero_grad(self) -> None:
        self.grad.zero_()

    # ------------------------------------------------------- whole-model sync
    @torch.no_grad()
    def refresh_shadow(self) -> None:
        """shadow <- master (after an external write to ``master``, e.g. a pull or a load)."""
        if self.shadow is not None:
            from ..ops import bucket as _b
            _b.scale_cast(self.master, self.shadow)

    @torch.no_grad()
    def load_flat(self, flat: torch.Tensor) -> None:
        self.master[: self.used].copy_(flat[: self.used].to(self.master.dtype))
        self.refresh_shadow()

    def nbytes(self) -> int:
        n = self.master.numel() * 4 + self.grad.numel() * self.grad.element_size()
        if self.shadow is not None:
            n += self.shadow.numel() * 2
        return n

    def checksum(self) -> torch.Tensor:
        """fp64 sum and sum of squares of ``master`` -- compared across ranks in debug mode
        to detect replica divergence."""
        m = self.master[: self.used].double()
        return torch.stack([m.sum(), (m * m).sum()])


@dataclass
class Bucket:
    index: int
    start: int            # element offset in the arenas (multiple of ALIGN * shard_ways)
    end: int
    slots: Sequence[Slot]

    @property
    def numel(self) -> int:
        return self.end - self.start


class BucketPlanner:
    """Cuts the flat index space into contiguous buckets of whole slots.

    ``cap_bytes`` bounds a bucket; ``first_cap_bytes`` lets bucket 0 be smaller so
    the first collective starts early in backward; ``last_cap_bytes`` cuts a small
    bucket off the end: the last bucket becomes ready only when backward is over, so
    its collective is the one nothing can hide, and everything before the cut (in
    ResNet-50 the parameter-heavy late stages, whose gradients exist early) then
    travels while the early layers' backward still runs. Slots are padded to
    ``ALIGN * shard_ways`` elements by :class:`FlatParams`, so every bucket splits
    evenly into ``shard_ways`` aligned shards for reduce-scatter / all-gather.

    On a fully connected xGMI node a collective is bound by the per-link rate,
    and each of the 7 peers receives 1/8 of a bucket: buckets of 32-64 MiB keep
    every per-link transfer in the multi-MiB range where link bandwidth, not
    launch latency, dominates.
    """

    def __init__(self, flat: FlatParams, cap_bytes: int = 48 << 20, first_cap_bytes: int | None = None,
                 last_cap_bytes: int | None = None):
        self.flat = flat
        self.shard_ways = flat.shard_ways
        esize = flat.grad.element_size()
        cap = max(1, cap_bytes // esize)
        limit = max(1, (first_cap_bytes or cap_bytes) // esize)
        buckets: list[Bucket] = []
        cur: list[Slot] = []
        start = 0
        for s in flat.slots:
            end = s.offset + _round_up(s.numel, flat.align)
            if cur and end - start > limit:
                buckets.append(Bucket(len(buckets), start, s.offset, tuple(cur)))
                start, cur, limit = s.offset, [], cap
            cur.append(s)
        if last_cap_bytes and len(cur) > 1 and flat.total - start > last_cap_bytes // esize:
            # the longest run of trailing slots that fits the small cap (at least one slot) becomes its own bucket
            cut = len(cur) - 1
            while cut > 1 and flat.total - cur[cut - 1].offset <= last_cap_bytes // esize:
                cut -= 1
            buckets.append(Bucket(len(buckets), start, cur[cut].offset, tuple(cur[:cut])))
            start, cur = cur[cut].offset, cur[cut:]
        buckets.append(Bucket(len(buckets), start, flat.total, tuple(cur)))
        self.buckets = buckets
        self.slot_bucket = {id(s.param): b for b in buckets for s in b.slots}

    def shard_range(self, b: Bucket, rank: int) -> tuple[int, int]:
        n = b.numel // self.shard_ways
        return b.start + rank * n, b.start + (rank + 1) * n
