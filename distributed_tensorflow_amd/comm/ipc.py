"""Peer-pointer collectives: hand-written all-reduce schedules on HIP IPC mapped buffers.

Every rank of a node maps the registered buffer (the gradient arena) of every other rank and
runs the kernels of ``csrc/kernels/p2p.hip`` on raw peer pointers. A schedule is a list of
*phases*; inside a phase every rank only reads regions no rank writes in that phase, and phases
are separated by a device-side barrier, so no host thread ever blocks and the whole collective
is stream-ordered (it can run on a side stream under the backward pass).

Schedules (``plan``):

``direct``   one-hop reduce-scatter + all-gather: rank r sums shard r of all peers, then pulls
             the other reduced shards; all 7 xGMI links of a GPU carry traffic at once
``oneshot``  every rank sums the whole buffer of all peers into scratch (one hop; small buffers)
``ring``     the classic 2(W-1)-step ring, kept for comparison: it is bound by ONE link
``tree``     binomial reduce to rank 0 + binomial broadcast: log2(W) hops for tiny buffers
``fused``    the ``direct`` schedule as ONE kernel launch per rank: a resident grid sized from the bucket runs entry
             barrier, reduce-scatter, barrier, all-gather and exit barrier itself (grid counters inside the device,
             growing sequence numbers in the peers' flag arrays), instead of five launches per bucket; the reduced
             shard is stored through the caches, so no barrier writes a cache back

The schedule is pure Python data, so it is verified on the CPU by an emulator
(:func:`emulate`) that also detects read/write races inside a phase; the kernels are verified
on one GPU with virtual ranks (:class:`LocalPeers`) and with several processes sharing the
device (:class:`PeerGroup`). The reference has no counterpart (its all-reduce is TensorFlow's
CPU ring, reference decentralized/native/dist_keras.py:77-78).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import torch

from ..ops import _lib

ALIGN_BYTES = 16


# ------------------------------------------------------------------------------- schedules
@dataclass(frozen=True)
class Op:
    """``dst_rank``'s buffer (or scratch) [off, off+n) = scale * sum of ``srcs`` ranks' buffers [off, off+n)."""
    kind: str            # 'reduce' | 'gather' | 'unscratch'
    srcs: tuple = ()     # source ranks, in summation order
    off: int = 0         # element offset inside the registered region
    n: int = 0           # elements
    scale: float = 1.0
    to_scratch: bool = False
    shard: int = 0       # gather: elements per shard


def _shards(n: int, world: int, esz: int) -> int:
    """Elements per shard: ceil(n / world) rounded up to 16 bytes."""
    q = ALIGN_BYTES // esz
    per = -(-n // world)
    return -(-per // q) * q


def plan(algorithm: str, rank: int, world: int, n: int, esz: int, scale: float = 1.0) -> list[list[Op]]:
    """Phases of ``rank`` for an all-reduce of ``n`` elements of ``esz`` bytes. A barrier runs
    before the first phase and after every phase."""
    if world == 1:
        return [[Op("reduce", (0,), 0, n, scale)]] if scale != 1.0 else []
    sh = _shards(n, world, esz)
    if algorithm == "direct":
        lo = min(rank * sh, n)
        cnt = max(0, min(sh, n - lo))
        p1 = [Op("reduce", tuple(range(world)), lo, cnt, scale)] if cnt else []
        return [p1, [Op("gather", shard=sh, n=n)]]
    if algorithm == "oneshot":
        return [[Op("reduce", tuple(range(world)), 0, n, scale, to_scratch=True)], [Op("unscratch", n=n)]]
    if algorithm == "ring":
        left = (rank - 1) % world
        phases: list[list[Op]] = []

        def chunk(c):
            lo = min(c * sh, n)
            return lo, max(0, min(sh, n - lo))

        for s in range(world - 1):                     # reduce-scatter: after step s rank r owns a partial of chunk r-s-1
            lo, cnt = chunk((rank - s - 1) % world)
            last = s == world - 2
            phases.append([Op("reduce", (rank, left), lo, cnt, scale if last else 1.0)] if cnt else [])
        for s in range(world - 1):                     # all-gather: pull the finished chunk r-s from the left
            lo, cnt = chunk((rank - s) % world)
            phases.append([Op("reduce", (left,), lo, cnt, 1.0)] if cnt else [])
        return phases
    if algorithm == "tree":
        phases = []
        d = 1
        dists = []
        while d < world:
            dists.append(d)
            d *= 2
        for d in dists:                                # reduce towards rank 0
            if rank % (2 * d) == 0 and rank + d < world:
                phases.append([Op("reduce", (rank, rank + d), 0, n, 1.0)])
            else:
                phases.append([])
        if scale != 1.0:
            phases.append([Op("reduce", (0,), 0, n, scale)] if rank == 0 else [])
        for d in reversed(dists):                      # broadcast back
            if rank % (2 * d) == d:
                phases.append([Op("reduce", (rank - d,), 0, n, 1.0)])
            else:
                phases.append([])
        return phases
    raise ValueError(f"unknown peer-pointer all-reduce algorithm {algorithm!r}")


def pick_algorithm(nbytes: int, world: int) -> str:
    """One hop and one barrier pair for small buffers; link-parallel direct schedule otherwise."""
    if world <= 2 or nbytes <= 512 * 1024:
        return "oneshot"
    return "direct"


def fused_grid(nbytes: int, world: int, cap: int = 256, max_threads: int = 256) -> tuple[int, int]:
    """(blocks, threads per block) of the single-launch all-reduce for a bucket of ``nbytes``: 32 blocks of 256
    threads up to 2 MiB (latency bound: three barriers, each as slow as its slowest block), then one more block per
    128 KiB that a rank moves (it reads ``world`` shards and pulls ``world - 1``), up to ``cap`` -- the CU budget
    the strategy grants a collective that runs under the backward pass (at most one block per CU: every block must be
    able to become resident) -- and from there wider blocks, up to ``max_threads``. 64 x 256 threads keep seven xGMI
    links busy; wide blocks only pay where the transfer is bound by device memory (ranks sharing one GPU)."""
    cap = max(1, min(int(cap), 256))
    want = -(-2 * nbytes // (128 << 10))
    blocks = max(min(32, cap), min(cap, want))
    threads = 256
    while threads < min(int(max_threads), 1024) and blocks * (threads // 256) < want:
        threads *= 2
    return blocks, threads


def emulate(algorithm: str, bufs: list, scale: float = 1.0) -> None:
    """Runs the schedule on host arrays (numpy or CPU tensors), in place, phase by phase, and
    raises if a phase reads a region that another rank writes in the same phase."""
    world, n = len(bufs), len(bufs[0])
    esz = bufs[0].element_size() if hasattr(bufs[0], "element_size") else bufs[0].itemsize
    plans = [plan(algorithm, r, world, n, esz, scale) for r in range(world)]
    scratch = [b.clone() if hasattr(b, "clone") else b.copy() for b in bufs]
    for ph in range(max((len(p) for p in plans), default=0)):
        writes, reads, results = [], [], []
        for r in range(world):
            for op in (plans[r][ph] if ph < len(plans[r]) else []):
                if op.kind == "reduce":
                    acc = None
                    for j in op.srcs:
                        seg = bufs[j][op.off:op.off + op.n]
                        seg = seg.double() if hasattr(seg, "double") else seg.astype("float64")
                        acc = seg if acc is None else acc + seg
                        if not (j == r and not op.to_scratch):
                            reads.append((j, op.off, op.off + op.n, r))
                    results.append((r, op.to_scratch, op.off, op.n, acc * op.scale))
                    if not op.to_scratch:
                        writes.append((r, op.off, op.off + op.n))
                elif op.kind == "gather":
                    for j in range(world):
                        lo, hi = min(j * op.shard, n), min((j + 1) * op.shard, n)
                        if j != r and hi > lo:
                            reads.append((j, lo, hi, r))
                            results.append((r, False, lo, hi - lo, bufs[j][lo:hi]))
                            writes.append((r, lo, hi))
                elif op.kind == "unscratch":
                    results.append((r, False, 0, op.n, scratch[r][:op.n]))
                    writes.append((r, 0, op.n))
        for (wr, wl, wh) in writes:
            for (rr, rl, rh, reader) in reads:
                if wr == rr and reader != wr and wl < rh and rl < wh:
                    raise AssertionError(f"{algorithm}: phase {ph}: rank {reader} reads [{rl},{rh}) of rank {rr} "
                                         f"while rank {wr} writes [{wl},{wh})")
        vals = [(r, s, off, cnt, (v.clone() if hasattr(v, "clone") else v.copy())) for r, s, off, cnt, v in results]
        for r, to_scratch, off, cnt, v in vals:
            tgt = scratch[r] if to_scratch else bufs[r]
            if hasattr(tgt, "dtype") and hasattr(v, "to"):
                tgt[off:off + cnt] = v.to(tgt.dtype)
            else:
                tgt[off:off + cnt] = v.astype(tgt.dtype)


# ------------------------------------------------------------------------------- native
def _k():
    lib = _lib.kernels()
    if not getattr(lib, "_p2p_declared", False):
        vp, ll, i = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
        lib.dtf_p2p_reduce.argtypes = [ctypes.POINTER(vp), i, vp, ll, ll, i, ctypes.c_float, i, vp]
        lib.dtf_p2p_gather.argtypes = [ctypes.POINTER(vp), i, i, vp, ll, ll, i, vp]
        lib.dtf_p2p_barrier.argtypes = [ctypes.POINTER(vp), i, i, ctypes.c_uint, i, vp, vp]
        lib.dtf_p2p_allreduce_fused.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), i, i, vp, ll, ll, i,
                                                ctypes.c_float, ctypes.c_uint, i, i, i, vp]
        lib.dtf_p2p_allreduce_fused_virtual.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), i, ctypes.POINTER(vp),
                                                        ll, ll, i, ctypes.c_float, ctypes.c_uint, i, i, i, vp]
        lib._p2p_declared = True
    return lib


def _rt():
    lib = _lib.runtime()
    if not getattr(lib, "_ipc_declared", False):
        vp, ull = ctypes.c_void_p, ctypes.c_ulonglong
        lib.dtf_ipc_export.argtypes = [vp, vp, ctypes.POINTER(ull), ctypes.POINTER(ull)]
        lib.dtf_ipc_open.argtypes = [vp, ull, ctypes.POINTER(vp)]
        lib.dtf_ipc_close.argtypes = [vp]
        lib.dtf_ipc_alloc_flags.argtypes = [ull, ctypes.POINTER(vp)]
        lib.dtf_ipc_free.argtypes = [vp]
        lib.dtf_ipc_read.argtypes = [vp, vp, ull]
        lib._ipc_declared = True
    return lib


def _ptr_array(ptrs: list[int]):
    arr = (ctypes.c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


class _Executor:
    """Launches the ops of one rank on raw pointers."""

    def __init__(self, rank: int, world: int, device, max_blocks: int = 0):
        self.rank, self.world, self.device, self.max_blocks = rank, world, torch.device(device), max_blocks

    def run_op(self, op: Op, ptrs: list[int], scratch_ptr: int, dtype: torch.dtype, base_off: int) -> None:
        """``ptrs[j]`` = address of element 0 of rank j's registered buffer; ``base_off`` = first
        element of the region being reduced."""
        lib, st = _k(), _lib.stream_ptr(self.device)
        esz = 4 if dtype == torch.float32 else 2
        code = _lib.dtype_code(dtype)
        if op.kind == "reduce":
            srcs = _ptr_array([ptrs[j] for j in op.srcs])
            off = base_off + op.off
            out = (scratch_ptr + op.off * esz) if op.to_scratch else (ptrs[self.rank] + off * esz)
            rc = lib.dtf_p2p_reduce(srcs, len(op.srcs), out, off, op.n, code, op.scale, self.max_blocks, st)
        elif op.kind == "gather":
            region = _ptr_array([p + base_off * esz for p in ptrs])
            rc = lib.dtf_p2p_gather(region, self.world, self.rank, ptrs[self.rank] + base_off * esz, op.shard * esz,
                                    op.n * esz, self.max_blocks, st)
        elif op.kind == "unscratch":
            srcs = _ptr_array([scratch_ptr])
            rc = lib.dtf_p2p_reduce(srcs, 1, ptrs[self.rank] + base_off * esz, 0, op.n, code, 1.0, self.max_blocks, st)
        else:  # pragma: no cover
            raise ValueError(op.kind)
        _lib.check(rc, f"p2p {op.kind}")


class LocalPeers:
    """Virtual ranks inside one process: the "peers" are W tensors on the same GPU. Phases run
    rank after rank on one stream, which is a valid barrier, so kernels and schedules are
    exercised without xGMI."""

    def __init__(self, bufs: list[torch.Tensor]):
        self.bufs = bufs
        self.world = len(bufs)
        self.scratch = [torch.empty_like(b) for b in bufs]

    def _fused(self, scale: float, start: int, n: int, blocks: int | None = None, budget_ms: int = 5000) -> None:
        """The single-launch all-reduce with the W virtual ranks as the rows of ONE grid, so that they are resident
        together by construction; each row runs the code of a real rank against its own counters, flag array and
        status word and synchronises with the others exactly as W processes would. ``blocks`` (per rank; default:
        sized from the region, W x blocks <= 256)."""
        b0 = self.bufs[0]
        dev = b0.device
        if not hasattr(self, "_fz"):
            rt = _rt()
            flags = []
            for _ in range(self.world):
                p = ctypes.c_void_p()
                _lib.check(rt.dtf_ipc_alloc_flags(4 * 64, ctypes.byref(p)), "flag allocation")
                flags.append(p.value)
            self._fz = dict(flags=flags, status=[torch.zeros(1, dtype=torch.int32, device=dev) for _ in self.bufs],
                            arrive=0)
        z = self._fz
        threads = 256
        if blocks is None:
            blocks, threads = fused_grid(n * b0.element_size(), self.world, cap=256 // self.world, max_threads=1024)
        rc = _k().dtf_p2p_allreduce_fused_virtual(
            _ptr_array([b.data_ptr() for b in self.bufs]), _ptr_array(z["flags"]), self.world,
            _ptr_array([t.data_ptr() for t in z["status"]]), start, n,
            _lib.dtype_code(b0.dtype), scale, z["arrive"], blocks, threads, budget_ms, _lib.stream_ptr(dev))
        _lib.check(rc, "p2p allreduce_fused (virtual ranks)")
        z["arrive"] = (z["arrive"] + 3 * blocks) & 0xFFFFFFFF

    def fused_status(self) -> list[int]:
        return [int(t.item()) for t in self._fz["status"]]

    def all_reduce(self, algorithm: str = "direct", scale: float = 1.0, start: int = 0, numel: int | None = None):
        b0 = self.bufs[0]
        n = b0.numel() - start if numel is None else numel
        esz = b0.element_size()
        if (start * esz) % ALIGN_BYTES:
            raise ValueError("region must start on a 16-byte boundary")
        if algorithm == "fused":
            return self._fused(scale, start, n)
        ptrs = [b.data_ptr() for b in self.bufs]
        plans = [plan(algorithm, r, self.world, n, esz, scale) for r in range(self.world)]
        ex = [_Executor(r, self.world, b0.device) for r in range(self.world)]
        for ph in range(max((len(p) for p in plans), default=0)):
            for r in range(self.world):
                for op in (plans[r][ph] if ph < len(plans[r]) else []):
                    ex[r].run_op(op, ptrs, self.scratch[r].data_ptr(), b0.dtype, start)


class PeerGroup:
    """One rank per process. ``comm`` is only used to exchange IPC handles (``gather_objects``:
    callable returning the list of every rank's object; default ``all_gather_object``)."""

    def __init__(self, comm, gather_objects=None, barrier_budget_ms: int = 20000, max_blocks: int = 0,
                 fused_threads: int = 256):
        self.comm = comm
        self.rank, self.world, self.device = comm.rank, comm.world, comm.device
        self._gather = gather_objects or self._default_gather
        self._exec = _Executor(self.rank, self.world, self.device, max_blocks)
        self._budget = barrier_budget_ms
        self._epoch = 0
        self._handles: list[bytes] = []
        rt = _rt()
        p = ctypes.c_void_p()
        _lib.check(rt.dtf_ipc_alloc_flags(4 * 64, ctypes.byref(p)), "flag allocation")
        self._flags_own = p.value
        self._flag_ptrs = self._map(self._flags_own)
        # the single-launch all-reduce counts in words of its own (16 words into each rank's flag array)
        self._fused_flag_ptrs = [q + 64 for q in self._flag_ptrs]
        # CU budget of the single-launch all-reduce (its grid grows with the bucket up to this many blocks; the rest
        # of the device stays with the backward pass)
        self.fused_blocks = max_blocks if 0 < max_blocks <= 256 else 64
        self.fused_threads = fused_threads      # widest block of that grid (1024 only pays on a shared GPU)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._registered: dict[int, tuple[list[int], torch.Tensor]] = {}

    def _default_gather(self, obj):
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.comm.group)
        return out

    def _map(self, own_ptr: int) -> list[int]:
        """Exports ``own_ptr``, gathers everyone's handle, maps the peers; returns pointers by rank."""
        rt = _rt()
        n = rt.dtf_ipc_handle_bytes()
        h = ctypes.create_string_buffer(n)
        off, size = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        _lib.check(rt.dtf_ipc_export(own_ptr, h, ctypes.byref(off), ctypes.byref(size)), "hipIpcGetMemHandle")
        everyone = self._gather((h.raw, off.value))
        ptrs = []
        for r, (hb, o) in enumerate(everyone):
            if r == self.rank:
                ptrs.append(own_ptr)
                continue
            p = ctypes.c_void_p()
            _lib.check(rt.dtf_ipc_open(hb, o, ctypes.byref(p)), "hipIpcOpenMemHandle")
            self._handles.append(hb)
            ptrs.append(p.value)
        return ptrs

    def register(self, t: torch.Tensor) -> int:
        """Collective: makes ``t`` (same numel/dtype on every rank) reachable by every peer.
        Returns a key for :meth:`all_reduce`. Scratch of the same size is allocated locally."""
        key = t.data_ptr()
        if key not in self._registered:
            self._registered[key] = (self._map(key), torch.empty_like(t))
        return key

    def barrier(self) -> None:
        self._epoch += 1
        rc = _k().dtf_p2p_barrier(_ptr_array(self._flag_ptrs), self.world, self.rank, self._epoch, self._budget,
                                  self.status.data_ptr(), _lib.stream_ptr(self.device))
        _lib.check(rc, "p2p barrier")

    def _raise_if(self, s: int) -> None:
        if s:
            missing = [j for j in range(self.world) if s >> j & 1]
            own = " (this rank's own grid among them: some of its blocks were not resident in time)" \
                if self.rank in missing else ""
            raise RuntimeError(f"peer-pointer barrier timed out waiting for ranks {missing}{own}: the buffers reduced "
                               "since then hold partial sums")

    def check(self) -> None:
        """Host check of the barrier status word (synchronises the current stream)."""
        self._raise_if(int(self.status.item()))

    def check_async(self) -> None:
        """The once-per-step form of :meth:`check` without a host wait: raises if the snapshot of the status word
        taken by the previous call shows a timed-out barrier, then enqueues a new snapshot (an asynchronous copy into
        pinned host memory) behind the work submitted so far. A late or dead peer is therefore reported at most
        one step after the barrier gave up on it."""
        snap = getattr(self, "_snap", None)
        if snap is None:
            self._snap = snap = (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
        else:
            host, ev = snap
            ev.synchronize()        # recorded a whole step ago: returns at once
            self._raise_if(int(host[0]))
        snap[0].copy_(self.status, non_blocking=True)
        snap[1].record()

    def all_reduce(self, t: torch.Tensor, algorithm: str = "auto", scale: float = 1.0, base: torch.Tensor | None = None):
        """All-reduce ``t``, a 16-byte aligned slice of the registered tensor ``base`` (default: ``t``
        itself is registered). Stream-ordered on the current stream."""
        base = t if base is None else base
        key = self.register(base)
        ptrs, scratch = self._registered[key]
        esz = t.element_size()
        start = (t.data_ptr() - base.data_ptr()) // esz
        if ((t.data_ptr() - base.data_ptr()) % ALIGN_BYTES) or not t.is_contiguous():
            raise ValueError("the reduced slice must be contiguous and start on a 16-byte boundary")
        if algorithm == "auto":
            algorithm = pick_algorithm(t.numel() * esz, self.world)
        if algorithm == "fused":
            if not hasattr(self, "_fused_arrive"):
                self._fused_arrive = 0
            blocks, threads = fused_grid(t.numel() * esz, self.world, cap=self.fused_blocks,
                                         max_threads=self.fused_threads)
            rc = _k().dtf_p2p_allreduce_fused(_ptr_array(ptrs), _ptr_array(self._fused_flag_ptrs), self.world, self.rank,
                                              self.status.data_ptr(), start, t.numel(),
                                              _lib.dtype_code(t.dtype), scale, self._fused_arrive,
                                              blocks, threads, self._budget, _lib.stream_ptr(self.device))
            _lib.check(rc, "p2p allreduce_fused")
            self._fused_arrive = (self._fused_arrive + 3 * blocks) & 0xFFFFFFFF
            return algorithm
        phases = plan(algorithm, self.rank, self.world, t.numel(), esz, scale)
        if self.world > 1:
            self.barrier()
        for ops in phases:
            for op in ops:
                self._exec.run_op(op, ptrs, scratch.data_ptr() + start * esz, t.dtype, start)
            if self.world > 1:
                self.barrier()
        return algorithm

    def close(self) -> None:
        rt = _rt()
        torch.cuda.synchronize(self.device)
        for hb in self._handles:
            rt.dtf_ipc_close(hb)
        self._handles.clear()
        if self._flags_own:
            rt.dtf_ipc_free(self._flags_own)
            self._flags_own = 0
