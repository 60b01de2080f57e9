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

``split_dtypes`` (the sharded parameter server's layout) places the bf16
parameters first and the fp32 ones (norm scales and offsets, biases kept in
fp32) after them, each group in reverse parameter order, and no bucket crosses
the boundary ``split``. An fp32 parameter IS its master copy, so that region
stays whole on every rank; below ``split`` the fp32 master is a copy that only
the optimizer reads, and :meth:`FlatParams.shard_master` cuts it down to the
ranges a rank owns (1/W of it): 4 of the 6 bytes a bf16 parameter costs in
arenas become 4/W.

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
                 shard_ways: int = 1, split_dtypes: bool = False):
        """``shard_ways``: every slot is padded to ``ALIGN * shard_ways`` elements so that any
        run of whole slots (a bucket) splits into ``shard_ways`` equal, aligned shards.
        ``split_dtypes``: bf16 parameters first, fp32 parameters after them (see the module text)."""
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
        order = list(reversed(named))
        self.split = None               # first element of the fp32 region (split layout only)
        if split_dtypes:
            order = [np_ for np_ in order if np_[1].dtype == torch.bfloat16] + \
                    [np_ for np_ in order if np_[1].dtype != torch.bfloat16]
        for n, p in order:
            if split_dtypes and self.split is None and p.dtype != torch.bfloat16:
                self.split = off
            self.slots.append(Slot(n, p, off, p.numel(), p.shape, p.dtype, _dense_strides(p)))
            off += _round_up(p.numel(), self.align)
        if split_dtypes and self.split is None:
            self.split = off
        self.used = off
        self.total = off
        self.master_base = 0            # arena coordinate of master[0]; > 0 once shard_master() has run
        self.master_shards = None       # (ranges, compact fp32 tensor) of the owned part below ``split``
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
        if s.offset < self.master_base:
            raise ValueError(f"the fp32 master of {s.name} is sharded across the ranks (shard_master)")
        return self.master.as_strided(tuple(s.shape), s.stride, s.offset - self.master_base)

    def m32(self, a: int, b: int) -> torch.Tensor:
        """``master`` in arena coordinates: the replicated fp32 region, or -- below ``split`` after
        :meth:`shard_master` -- the compact storage of a range this rank owns."""
        if a >= self.master_base:
            return self.master[a - self.master_base:b - self.master_base]
        ranges, store = self.master_shards
        base = 0
        for ra, rb in ranges:
            if ra <= a and b <= rb:
                return store[base + a - ra:base + b - ra]
            base += rb - ra
        raise ValueError(f"range [{a}, {b}) of the fp32 master is not owned by this rank")

    @torch.no_grad()
    def shard_master(self, ranges: Sequence[tuple[int, int]]) -> None:
        """Keeps, of the fp32 master below ``split`` (the copies of bf16 parameters), only ``ranges`` -- the
        shards this rank owns as a parameter server -- in compact storage, and releases the rest. The fp32
        region at and above ``split`` stays whole (those parameters are views of it). Afterwards ``master`` is
        that region only (``master_base == split``); use :meth:`m32` for arena coordinates."""
        if self.split is None:
            raise ValueError("shard_master needs the split layout (FlatParams(split_dtypes=True))")
        if self.master_base:
            raise ValueError("the master arena is already sharded")
        own = [(int(a), int(b)) for a, b in ranges if b > a and a < self.split]
        if any(b > self.split for _, b in own):
            raise ValueError("an owned range crosses the boundary between bf16 and fp32 parameters")
        store = torch.cat([self.master[a:b] for a, b in own]) if own else self.master.new_empty(0)
        hi = self.master[self.split:].clone()
        for s in self.slots:
            if s.offset >= self.split:
                s.param.data = hi.as_strided(tuple(s.shape), s.stride, s.offset - self.split)
            elif hasattr(s.param, "_dtf_master"):
                del s.param._dtf_master      # no full-precision copy of this parameter on this rank any more
        self.master = hi
        self.master_base = self.split
        self.master_shards = (own, store)

    @torch.no_grad()
    def write_master(self, full: torch.Tensor) -> None:
        """Loads ``full`` (fp32, the whole index space) into whatever part of the master this rank holds, and the
        bf16 working copies."""
        full = full.to(self.device)
        if not self.master_base:
            self.master[: self.used].copy_(full[: self.used])
            self.refresh_shadow()
            return
        self.master.copy_(full[self.master_base:self.total])
        ranges, store = self.master_shards
        base = 0
        for a, b in ranges:
            store[base:base + b - a].copy_(full[a:b])
            base += b - a
        from ..ops import bucket as _b
        _b.scale_cast(full[: self.master_base].contiguous(), self.shadow[: self.master_base])

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
        if self.shadow is not None and not self.master_base:
            # (with a sharded master the bf16 arena is itself the replicated working copy: the owners write their
            # shards of it in the optimizer kernel and the pull all-gathers it)
            from ..ops import bucket as _b
            _b.scale_cast(self.master, self.shadow)

    @torch.no_grad()
    def load_flat(self, flat: torch.Tensor) -> None:
        self.master[: self.used].copy_(flat[: self.used].to(self.master.dtype))
        self.refresh_shadow()

    def nbytes(self) -> int:
        n = self.master.numel() * 4 + self.grad.numel() * self.grad.element_size()
        if self.master_shards is not None:
            n += self.master_shards[1].numel() * 4
        if self.shadow is not None:
            n += self.shadow.numel() * 2
        return n

    def checksum(self) -> torch.Tensor:
        """fp64 sum and sum of squares of ``master`` -- compared across ranks in debug mode
        to detect replica divergence."""
        if self.master_base:
            # replicated state: the bf16 working copy below the boundary, the fp32 parameters above it
            lo, hi = self.shadow[: self.master_base].double(), self.master.double()
            return torch.stack([lo.sum() + hi.sum(), (lo * lo).sum() + (hi * hi).sum()])
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
        split = getattr(flat, "split", None)
        for s in flat.slots:
            end = s.offset + _round_up(s.numel, flat.align)
            if cur and (end - start > limit or s.offset == split):     # no bucket holds both kinds of parameter
                buckets.append(Bucket(len(buckets), start, s.offset, tuple(cur)))
                start, cur, limit = s.offset, [], cap
            cur.append(s)
        if last_cap_bytes and len(cur) > 1 and flat.total - start > last_cap_bytes // esize \
                and not (split is not None and start < split < flat.total):
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
