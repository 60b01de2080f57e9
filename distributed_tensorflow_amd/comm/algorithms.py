"""Hand-written all-reduce algorithms on point-to-point transfers.

The reference names a ``custom`` decentralized strategy but leaves it
unimplemented (reference initializer.py:179-181) and delegates the ``keras``
one to a CPU ring over gRPC (decentralized/native/dist_keras.py:77-78). These
are the algorithms behind this framework's ``custom`` strategy. They are built
from grouped send/recv (:meth:`Comm.exchange`, one RCCL group call per step) and
the n-ary reduction kernel, and they are chosen for a node whose 8 GPUs are
fully connected by point-to-point xGMI links:

``direct``
    one-hop reduce-scatter + all-gather. Rank r owns shard r; every rank sends
    shard j straight to rank j over its dedicated link, reduces the 7 incoming
    shards in ONE kernel pass, then sends the result to all peers. All 7 links
    of every GPU carry traffic in both phases; time ~ 2 * (S/8) / link_bw.
``ring``
    classic reduce-scatter ring + all-gather ring. A single ring drives one
    link per direction per GPU, so it is per-link bound: time ~ 2 * (7/8) * S /
    link_bw, about 7x the direct schedule on a full mesh.
``multiring``
    the buffer is split over edge-disjoint Hamiltonian rings of the complete
    graph (Walecki decomposition, both directions): 6 of the 7 links of every GPU
    are busy while each transfer stays a ring hop.
``tree``
    binomial-tree reduce to rank 0 then broadcast; log2(W) hops for small,
    latency-bound buffers.
``oneshot``
    every rank sends its whole buffer to every peer and reduces locally; one
    hop, for buffers of a few hundred KB and less.

All functions reduce ``buf`` in place to the SUM over ranks. ``buf.numel()``
must be a multiple of ``8 * world`` (the arenas guarantee more).
"""
from __future__ import annotations

from typing import Sequence

import torch

from ..ops import bucket as bops
from .backend import Comm


# Host-issued operations since the last reset: grouped transfers and reduction kernels (bench.py reports them per
# bucket -- the direct schedule costs 3 whatever the world size, a ring 3 (W - 1)).
stats = {"exchanges": 0, "kernels": 0}


def _wait(works) -> None:
    for w in works or ():
        w.wait()


def _exchange(comm: Comm, sends, recvs):
    stats["exchanges"] += 1
    return comm.exchange(sends, recvs)


def _axpy(dst, srcs, weights) -> None:
    stats["kernels"] += 1
    bops.nary_axpy(dst, srcs, weights)


def _chunks(buf: torch.Tensor, parts: int) -> list[torch.Tensor]:
    n = buf.numel()
    if n % parts:
        raise ValueError(f"buffer of {n} elements does not split into {parts} equal parts")
    c = n // parts
    return [buf[i * c:(i + 1) * c] for i in range(parts)]


def _scratch(buf: torch.Tensor, numel: int, scratch: torch.Tensor | None) -> torch.Tensor:
    if scratch is not None and scratch.numel() >= numel and scratch.dtype == buf.dtype:
        return scratch[:numel]
    return torch.empty(numel, dtype=buf.dtype, device=buf.device)


# ---------------------------------------------------------------------- direct
def all_reduce_direct(comm: Comm, buf: torch.Tensor, scratch: torch.Tensor | None = None) -> None:
    W, r = comm.world, comm.rank
    if W == 1:
        return
    ch = _chunks(buf, W)
    c = ch[0].numel()
    tmp = _scratch(buf, (W - 1) * c, scratch)
    peers = [p for p in range(W) if p != r]
    slots = {p: tmp[i * c:(i + 1) * c] for i, p in enumerate(peers)}
    _wait(_exchange(comm, [(ch[p], p) for p in peers], [(slots[p], p) for p in peers]))
    _axpy(ch[r], [slots[p] for p in peers], [1.0] * (W - 1))
    _wait(_exchange(comm, [(ch[r], p) for p in peers], [(ch[p], p) for p in peers]))


def reduce_scatter_direct(comm: Comm, buf: torch.Tensor, scratch: torch.Tensor | None = None) -> torch.Tensor:
    """First half of :func:`all_reduce_direct`; returns the reduced local shard (a view of buf)."""
    W, r = comm.world, comm.rank
    ch = _chunks(buf, W)
    if W == 1:
        return ch[0]
    c = ch[0].numel()
    tmp = _scratch(buf, (W - 1) * c, scratch)
    peers = [p for p in range(W) if p != r]
    slots = {p: tmp[i * c:(i + 1) * c] for i, p in enumerate(peers)}
    _wait(_exchange(comm, [(ch[p], p) for p in peers], [(slots[p], p) for p in peers]))
    _axpy(ch[r], [slots[p] for p in peers], [1.0] * (W - 1))
    return ch[r]


def all_gather_direct(comm: Comm, buf: torch.Tensor) -> None:
    """Second half of :func:`all_reduce_direct`: shard r of buf is valid on rank r; fill the rest."""
    W, r = comm.world, comm.rank
    if W == 1:
        return
    ch = _chunks(buf, W)
    peers = [p for p in range(W) if p != r]
    _wait(_exchange(comm, [(ch[r], p) for p in peers], [(ch[p], p) for p in peers]))


# ------------------------------------------------------------------------ ring
_ring_plans: dict = {}


def _ring_pass(comm: Comm, parts: Sequence[torch.Tensor], orders: Sequence[Sequence[int]],
               scratch: torch.Tensor | None) -> None:
    """Ring all-reduce of parts[k] over ring orders[k], all rings advancing in lock step."""
    W = comm.world
    # the send / receive lists of every step are a function of (buffers, rings): built once per bucket and reused
    # by every later step of the training run (the arenas never move)
    key = (id(comm), tuple((p.data_ptr(), p.numel()) for p in parts), tuple(tuple(o) for o in orders),
           None if scratch is None else scratch.data_ptr())
    plan = _ring_plans.get(key)
    if plan is None:
        pos = [list(o).index(comm.rank) for o in orders]
        nxt = [o[(p + 1) % W] for o, p in zip(orders, pos)]
        prv = [o[(p - 1) % W] for o, p in zip(orders, pos)]
        chs = [_chunks(p, W) for p in parts]
        csz = [c[0].numel() for c in chs]
        tmp = _scratch(parts[0], sum(csz), scratch)
        tmps, o = [], 0
        for c in csz:
            tmps.append(tmp[o:o + c]); o += c
        K = range(len(parts))
        rs = [([(chs[k][(pos[k] - step) % W], nxt[k]) for k in K], [(tmps[k], prv[k]) for k in K],
               [(chs[k][(pos[k] - step - 1) % W], tmps[k]) for k in K]) for step in range(W - 1)]
        ag = [([(chs[k][(pos[k] + 1 - step) % W], nxt[k]) for k in K],
               [(chs[k][(pos[k] - step) % W], prv[k]) for k in K]) for step in range(W - 1)]
        plan = (rs, ag, tmp)
        if scratch is not None:               # (a temporary scratch buffer would be freed under a cached plan)
            if len(_ring_plans) > 256:
                _ring_plans.clear()
            _ring_plans[key] = plan
    rs, ag, _ = plan
    for sends, recvs, sums in rs:   # reduce-scatter
        _wait(_exchange(comm, sends, recvs))
        for dst, src in sums:
            _axpy(dst, [src], [1.0])
    for sends, recvs in ag:         # all-gather
        _wait(_exchange(comm, sends, recvs))


def all_reduce_ring(comm: Comm, buf: torch.Tensor, scratch: torch.Tensor | None = None) -> None:
    if comm.world == 1:
        return
    _ring_pass(comm, [buf], [list(range(comm.world))], scratch)


def hamiltonian_rings(world: int) -> list[list[int]]:
    """Directed Hamiltonian rings of the complete graph on ``world`` vertices that share no
    directed edge. For even ``world`` (Walecki): (world-2)/2 undirected cycles, each used in both
    directions -> world-2 rings (6 for 8 GPUs: 6 of the 7 links per GPU). For odd ``world``:
    (world-1)/2 cycles -> world-1 rings."""
    if world < 3:
        return [list(range(world))]
    cycles: list[list[int]] = []
    if world % 2 == 0:
        n = world - 1  # vertices 0..n-1 on a circle, vertex n is the hub
        for i in range((world - 2) // 2):
            zig = [i]
            for k in range(1, n):
                step = (k + 1) // 2
                zig.append((i + step) % n if k % 2 else (i - step) % n)
            cycles.append([n] + zig)
    else:
        n = world - 1  # Walecki on K_{n+1} with the hub, for odd world: hub = n
        m = n // 2
        for i in range(m):
            zig = [i]
            for k in range(1, n):
                step = (k + 1) // 2
                zig.append((i + step) % n if k % 2 else (i - step) % n)
            cycles.append([n] + zig)
    rings = []
    for c in cycles:
        rings.append(c)
        rings.append([c[0]] + c[:0:-1])
    return rings


def all_reduce_multiring(comm: Comm, buf: torch.Tensor, scratch: torch.Tensor | None = None) -> None:
    W = comm.world
    if W == 1:
        return
    rings = hamiltonian_rings(W)
    n = buf.numel()
    quantum = 8 * W
    per = (n // len(rings)) // quantum * quantum
    if per == 0:
        return all_reduce_ring(comm, buf, scratch)
    parts, orders, o = [], [], 0
    for k, ring in enumerate(rings):
        end = n if k == len(rings) - 1 else o + per
        parts.append(buf[o:end]); orders.append(ring); o = end
    _ring_pass(comm, parts, orders, scratch)


# ------------------------------------------------------------------------ tree
def all_reduce_tree(comm: Comm, buf: torch.Tensor, scratch: torch.Tensor | None = None) -> None:
    W, r = comm.world, comm.rank
    if W == 1:
        return
    tmp = _scratch(buf, buf.numel(), scratch)
    d = 1
    while d < W:  # reduce towards rank 0
        if r % (2 * d) == d:
            _wait(_exchange(comm, [(buf, r - d)], []))
        elif r % (2 * d) == 0 and r + d < W:
            _wait(_exchange(comm, [], [(tmp, r + d)]))
            _axpy(buf, [tmp], [1.0])
        d *= 2
    d //= 2
    while d >= 1:  # broadcast back down
        if r % (2 * d) == 0 and r + d < W:
            _wait(_exchange(comm, [(buf, r + d)], []))
        elif r % (2 * d) == d:
            _wait(_exchange(comm, [], [(buf, r - d)]))
        d //= 2


# --------------------------------------------------------------------- oneshot
def all_reduce_oneshot(comm: Comm, buf: torch.Tensor, scratch: torch.Tensor | None = None) -> None:
    W, r = comm.world, comm.rank
    if W == 1:
        return
    n = buf.numel()
    tmp = _scratch(buf, (W - 1) * n, scratch)
    peers = [p for p in range(W) if p != r]
    slots = [tmp[i * n:(i + 1) * n] for i in range(W - 1)]
    _wait(_exchange(comm, [(buf, p) for p in peers], [(s, p) for s, p in zip(slots, peers)]))
    # fixed summation order (by rank) so every replica computes bit-identical sums
    order = sorted(range(W))
    srcs = {p: s for p, s in zip(peers, slots)}
    if order[0] == r:
        _axpy(buf, [srcs[p] for p in order[1:]], [1.0] * (W - 1))
    else:
        first = srcs[order[0]]
        rest = [buf if p == r else srcs[p] for p in order[1:]]
        _axpy(first, rest, [1.0] * (W - 1))
        buf.copy_(first)


ALGORITHMS = {
    "direct": all_reduce_direct,
    "ring": all_reduce_ring,
    "multiring": all_reduce_multiring,
    "tree": all_reduce_tree,
    "oneshot": all_reduce_oneshot,
}


def choose_algorithm(nbytes: int, world: int) -> str:
    """Size-based default: one hop for tiny buffers, tree for small ones, the direct
    reduce-scatter + all-gather schedule once per-link transfers reach ~100 KB."""
    if world <= 2 or nbytes <= 256 << 10:
        return "oneshot"
    if nbytes <= 1 << 20:
        return "tree"
    return "direct"


def all_reduce(comm: Comm, buf: torch.Tensor, algo: str = "auto", scratch: torch.Tensor | None = None) -> str:
    if algo == "auto":
        algo = choose_algorithm(buf.numel() * buf.element_size(), comm.world)
    if buf.numel() % (8 * comm.world) and algo in ("direct", "ring", "multiring"):
        algo = "oneshot" if buf.numel() * buf.element_size() <= 1 << 20 else "tree"
    ALGORITHMS[algo](comm, buf, scratch)
    return algo
