"""Direct RCCL communicator (``csrc/runtime/rccl_comm.cpp``) behind the :class:`Comm` interface.

``Comm`` drives RCCL through ``torch.distributed``; this class drives it through the
framework's own binding: the unique id travels over the framed-TCP control plane (the one
role left for the reference's socket layer, reference centralized/network.py:4-28), every
collective is enqueued on a dedicated communication stream ordered against the compute stream
with events, and asynchronous errors are polled so that a dead peer ends the job instead of
hanging it (the reference waits for ever, centralized/server.py:95).

Select it with ``DTF_COMM=rccl`` or ``Comm.init(..., backend="rccl")``.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from ..ops import _lib
from . import tcp
from .backend import Comm

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.int32: 3, torch.int64: 4, torch.uint8: 5,
       torch.float64: 6}
_OP = {dist.ReduceOp.SUM: 0, dist.ReduceOp.PRODUCT: 1, dist.ReduceOp.MAX: 2, dist.ReduceOp.MIN: 3}
_KIND_ID = 0x52434944  # 'RCID'


class RcclError(RuntimeError):
    pass


def _rt():
    lib = _lib.runtime()
    if not getattr(lib, "_rccl_declared", False):
        vp, ull, i = ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_int
        lib.dtf_rccl_unique_id.argtypes = [vp, i]
        lib.dtf_rccl_init.argtypes = [vp, i, i, i, ctypes.POINTER(i)]
        lib.dtf_rccl_init.restype = vp
        lib.dtf_rccl_split.argtypes = [vp, i, i, ctypes.POINTER(vp)]
        lib.dtf_rccl_all_reduce.argtypes = [vp, vp, vp, ull, i, i, vp]
        lib.dtf_rccl_reduce_scatter.argtypes = [vp, vp, vp, ull, i, i, vp]
        lib.dtf_rccl_all_gather.argtypes = [vp, vp, vp, ull, i, vp]
        lib.dtf_rccl_broadcast.argtypes = [vp, vp, vp, ull, i, i, vp]
        lib.dtf_rccl_reduce.argtypes = [vp, vp, vp, ull, i, i, i, vp]
        lib.dtf_rccl_all_to_all.argtypes = [vp, vp, vp, ull, i, vp]
        lib.dtf_rccl_send.argtypes = [vp, vp, ull, i, i, vp]
        lib.dtf_rccl_recv.argtypes = [vp, vp, ull, i, i, vp]
        lib.dtf_rccl_async_error.argtypes = [vp, ctypes.POINTER(i)]
        lib.dtf_rccl_abort.argtypes = [vp]
        lib.dtf_rccl_destroy.argtypes = [vp]
        lib.dtf_rccl_error_string.argtypes = [i]
        lib.dtf_rccl_error_string.restype = ctypes.c_char_p
        lib._rccl_declared = True
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RcclError(f"{what}: {_rt().dtf_rccl_error_string(rc).decode()} ({rc})")


class _Work:
    """Handle of a collective enqueued on the communication stream."""

    def __init__(self, done: "torch.cuda.Event", device):
        self.done, self.device = done, device

    def wait(self) -> None:
        torch.cuda.current_stream(self.device).wait_event(self.done)

    def is_completed(self) -> bool:
        return self.done.query()


def exchange_unique_id(rank: int, world: int, address: str, port: int, timeout_s: float = 120.0) -> bytes:
    """Rank 0 creates the id and serves it to the ``world - 1`` other ranks over framed TCP."""
    lib = _rt()
    n = lib.dtf_rccl_unique_id_bytes()
    if rank == 0:
        buf = ctypes.create_string_buffer(n)
        _check(lib.dtf_rccl_unique_id(buf, n), "ncclGetUniqueId")
        payload = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
        lst = tcp.Listener(address, port)
        conns = []
        try:
            for _ in range(world - 1):
                c = lst.accept(timeout_s, timeout_s)
                conns.append(c)
                c.send(_KIND_ID, {"bytes": n}, payload)
        finally:
            for c in conns:
                c.close()
            lst.close()
        return buf.raw
    c = tcp.connect(address, port, timeout=timeout_s, conn_timeout=timeout_s)
    try:
        kind, meta, payload = c.recv()
    finally:
        c.close()
    if kind != _KIND_ID or payload is None or payload.numel() != n:
        raise RcclError("malformed unique-id message")
    return bytes(payload.numpy().tobytes())


class RcclComm(Comm):
    """``Comm`` on the native RCCL binding. Collectives are stream-ordered; ``async_op=True``
    returns a handle whose ``wait()`` makes the *current stream* wait (no host block)."""

    def __init__(self, rank: int, world: int, device, handle, uid: bytes | None = None):
        super().__init__(rank, world, device, group=None)
        self._h = ctypes.c_void_p(handle)
        self._stream = torch.cuda.Stream(self.device)
        self._uid = uid

    # ------------------------------------------------------------------ setup
    @classmethod
    def create(cls, rank: int, world: int, device, master_addr: str = "127.0.0.1", master_port: int = 29500,
               timeout_s: float = 120.0) -> "RcclComm":
        device = torch.device(device)
        if device.type != "cuda":
            raise ValueError("the RCCL communicator needs a GPU device")
        torch.cuda.set_device(device)
        uid = exchange_unique_id(rank, world, master_addr, master_port, timeout_s) if world > 1 else None
        lib = _rt()
        if uid is None:
            n = lib.dtf_rccl_unique_id_bytes()
            buf = ctypes.create_string_buffer(n)
            _check(lib.dtf_rccl_unique_id(buf, n), "ncclGetUniqueId")
            uid = buf.raw
        rc = ctypes.c_int(0)
        h = lib.dtf_rccl_init(uid, rank, world, device.index or 0, ctypes.byref(rc))
        if not h:
            _check(rc.value or -1, "ncclCommInitRank")
        return cls(rank, world, device, h, uid)

    def subgroup(self, ranks) -> "RcclComm":
        """A communicator over ``ranks`` by ``ncclCommSplit`` (collective on this communicator: every rank calls it,
        members and non-members alike, in the same order). A non-member gets a placeholder with rank -1, as
        :meth:`Comm.subgroup` does; the asynchronous parameter server builds its per-pair communicators this way."""
        ranks = list(ranks)
        member = self.rank in ranks
        out = ctypes.c_void_p(0)
        _check(_rt().dtf_rccl_split(self._h, 0 if member else -1, ranks.index(self.rank) if member else 0,
                                    ctypes.byref(out)), "ncclCommSplit")
        if not member:
            c = RcclComm(-1, len(ranks), self.device, 0)
        else:
            c = RcclComm(ranks.index(self.rank), len(ranks), self.device, out.value)
        c._global_ranks = ranks
        c.backend = "nccl"
        return c

    def async_error(self):
        """Error string of an asynchronous RCCL failure, or None (polled by ``comm.watchdog.Watchdog``)."""
        if not self._h:
            return None
        err = ctypes.c_int(0)
        if _rt().dtf_rccl_async_error(self._h, ctypes.byref(err)) != 0:
            return "ncclCommGetAsyncError failed"
        return _rt().dtf_rccl_error_string(err.value).decode() if err.value != 0 else None

    def shutdown(self) -> None:
        if self._h:
            torch.cuda.synchronize(self.device)
            _rt().dtf_rccl_destroy(self._h)
            self._h = ctypes.c_void_p(0)

    def abort(self) -> None:
        if self._h:
            _rt().dtf_rccl_abort(self._h)
            self._h = ctypes.c_void_p(0)

    def check_async_error(self) -> None:
        """Raises if RCCL has recorded an asynchronous failure (dead peer, broken link)."""
        err = ctypes.c_int(0)
        _check(_rt().dtf_rccl_async_error(self._h, ctypes.byref(err)), "ncclCommGetAsyncError")
        if err.value != 0:
            raise RcclError(f"asynchronous RCCL error: {_rt().dtf_rccl_error_string(err.value).decode()}")

    # ---------------------------------------------------------------- plumbing
    def _issue(self, fn, async_op: bool, *tensors):
        """Runs ``fn(stream_ptr)`` on the communication stream after everything already queued on
        the current stream; synchronous calls make the current stream wait for the result."""
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        self._stream.wait_event(ready)
        _check(fn(ctypes.c_void_p(self._stream.cuda_stream)), "rccl enqueue")
        done = torch.cuda.Event()
        done.record(self._stream)
        # The tensors stay alive until the communication stream is done with them by REFERENCE (dropped here once
        # their event has completed), not by record_stream: that makes the caching allocator poll an event per
        # recorded block on every allocation, which cost a 100 GB training step a factor of five.
        self._held = [h for h in getattr(self, "_held", ()) if not h[0].query()]
        self._held.append((done, tensors))
        w = _Work(done, self.device)
        if async_op:
            return w
        w.wait()
        return None

    @staticmethod
    def _dt(t: torch.Tensor) -> int:
        try:
            return _DT[t.dtype]
        except KeyError:
            raise TypeError(f"dtype {t.dtype} is not supported by the RCCL binding") from None

    # ------------------------------------------------------------ collectives
    def all_reduce(self, t, op=dist.ReduceOp.SUM, async_op=False):
        lib, p, n, d, o = _rt(), t.data_ptr(), t.numel(), self._dt(t), _OP[op]
        return self._issue(lambda s: lib.dtf_rccl_all_reduce(self._h, p, p, n, d, o, s), async_op, t)

    def reduce_scatter(self, out, inp, async_op=False):
        lib, d = _rt(), self._dt(inp)
        return self._issue(lambda s: lib.dtf_rccl_reduce_scatter(self._h, inp.data_ptr(), out.data_ptr(), out.numel(),
                                                                 d, 0, s), async_op, out, inp)

    def all_gather(self, out, inp, async_op=False):
        lib, d = _rt(), self._dt(inp)
        return self._issue(lambda s: lib.dtf_rccl_all_gather(self._h, inp.data_ptr(), out.data_ptr(), inp.numel(), d,
                                                             s), async_op, out, inp)

    def all_to_all(self, out, inp, async_op=False):
        lib, d = _rt(), self._dt(inp)
        return self._issue(lambda s: lib.dtf_rccl_all_to_all(self._h, inp.data_ptr(), out.data_ptr(),
                                                             inp.numel() // self.world, d, s), async_op, out, inp)

    def broadcast(self, t, src=0, async_op=False):
        lib, d = _rt(), self._dt(t)
        return self._issue(lambda s: lib.dtf_rccl_broadcast(self._h, t.data_ptr(), t.data_ptr(), t.numel(), d, src, s),
                           async_op, t)

    def reduce(self, t, dst=0, async_op=False):
        lib, d = _rt(), self._dt(t)
        return self._issue(lambda s: lib.dtf_rccl_reduce(self._h, t.data_ptr(), t.data_ptr(), t.numel(), d, 0, dst, s),
                           async_op, t)

    def barrier(self) -> None:
        t = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.all_reduce(t)
        torch.cuda.current_stream(self.device).synchronize()
        self.check_async_error()

    # -------------------------------------------------------------------- p2p
    def send(self, t, dst):
        lib, d = _rt(), self._dt(t)
        self._issue(lambda s: lib.dtf_rccl_send(self._h, t.data_ptr(), t.numel(), d, dst, s), False, t)

    def recv(self, t, src):
        lib, d = _rt(), self._dt(t)
        self._issue(lambda s: lib.dtf_rccl_recv(self._h, t.data_ptr(), t.numel(), d, src, s), False, t)

    def exchange(self, sends, recvs):
        """All sends and receives as one RCCL group: concurrent transfers, one per xGMI link."""
        if not sends and not recvs:
            return []
        lib = _rt()

        def go(s):
            rc = lib.dtf_rccl_group_start()
            for t, dst in sends:
                rc = rc or lib.dtf_rccl_send(self._h, t.data_ptr(), t.numel(), self._dt(t), dst, s)
            for t, src in recvs:
                rc = rc or lib.dtf_rccl_recv(self._h, t.data_ptr(), t.numel(), self._dt(t), src, s)
            rc2 = lib.dtf_rccl_group_end()
            return rc or rc2

        return [self._issue(go, True, *[t for t, _ in sends], *[t for t, _ in recvs])]

    def all_reduce_scalar(self, v, op=dist.ReduceOp.SUM):
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float32, device=self.device)
        self.all_reduce(t, op=op)
        return float(t)
