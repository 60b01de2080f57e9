"""Process-group layer: one process per GPU on ``torch.distributed``.

On a GPU node the backend is ``"nccl"``, which on ROCm is RCCL over xGMI; in the
CPU plumbing mode it is ``"gloo"`` over loopback, the direct analogue of the
reference's habit of running every role as a local process on 127.0.0.1
(reference initializer.py:134-145,164-173; centralized/server.py:160-163).

:class:`Comm` is a thin, explicit wrapper: every collective the strategies use
goes through it, world-size-1 runs short-circuit without a process group, and
every blocking call is bounded by the group timeout so a dead peer surfaces as
an error instead of the reference's indefinite ``barrier.wait()``
(centralized/server.py:95).
"""
from __future__ import annotations

import datetime
import os
from typing import Sequence

import torch
import torch.distributed as dist


class _StagedWork:
    """Completion handle of point-to-point transfers that were staged through host memory: ``wait`` finishes the
    transfers and copies what was received into the device tensors."""

    def __init__(self, works, copies):
        self.works, self.copies = works, copies

    def wait(self) -> None:
        for w in self.works:
            w.wait()
        for dst, host in self.copies:
            dst.copy_(host)
        self.works, self.copies = [], []


def _published_port(cell, rank: int, port: int | None, timeout_s: float) -> int:
    """Rank 0 (with ``port``) writes the port it bound into the shared cell; everyone else polls for it. With the
    framework's own RCCL binding, whose rendezvous listens on a given port, rank 0 reserves one by binding and
    keeps the socket until its listener is up (``SO_REUSEADDR`` lets the listener take it over)."""
    import time
    if rank == 0:
        if port is None:
            import socket
            s = socket.socket()
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            s.bind(("", 0))
            port = s.getsockname()[1]
            _held_sockets.append(s)
        cell.value = int(port)
        return int(port)
    t_end = time.time() + timeout_s
    while cell.value == 0:
        if time.time() > t_end:
            raise TimeoutError("rank 0 did not publish the rendezvous port")
        time.sleep(0.01)
    return int(cell.value)


_held_sockets: list = []


class Comm:
    def __init__(self, rank: int = 0, world: int = 1, device: torch.device | str = "cpu", group=None,
                 backend: str | None = None):
        self.rank, self.world = rank, world
        self.device = torch.device(device)
        self.group = group
        self.backend = backend or ("nccl" if self.device.type == "cuda" else "gloo")
        # gloo lacks reduce_scatter / all_to_all: those collectives are composed from the ones it has
        # (also when gloo carries GPU tensors, e.g. several test ranks sharing one device)
        self.composed = self.backend == "gloo"

    # ------------------------------------------------------------------ setup
    @classmethod
    def init(cls, rank: int, world: int, device: torch.device | str, master_addr: str = "127.0.0.1",
             master_port: int = 29500, timeout_s: float = 300.0, backend: str | None = None,
             port_cell=None) -> "Comm":
        """``port_cell`` (a ``multiprocessing.Value('i')`` shared by the ranks of one launcher, initially 0):
        rank 0 binds the rendezvous store to a port the operating system picks and publishes it there; the other
        ranks wait for it. No port is chosen before it is bound, so two jobs on one host cannot collide on it."""
        device = torch.device(device)
        if world == 1:
            return cls(0, 1, device)
        if backend is None:
            backend = os.environ.get("DTF_COMM") or ("nccl" if device.type == "cuda" else "gloo")
        if port_cell is not None and backend == "rccl":
            master_port = _published_port(port_cell, rank, None, timeout_s)
        if backend == "rccl":       # the framework's own RCCL binding instead of torch.distributed
            from .rccl import RcclComm
            return RcclComm.create(rank, world, device, master_addr, master_port, timeout_s)
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = device
            to = datetime.timedelta(seconds=timeout_s)
            if port_cell is not None:
                if rank == 0:
                    store = dist.TCPStore(master_addr, 0, world, True, to, wait_for_workers=False)
                    _published_port(port_cell, 0, store.port, timeout_s)
                else:
                    store = dist.TCPStore(master_addr, _published_port(port_cell, rank, None, timeout_s), world,
                                          False, to)
                dist.init_process_group(backend, store=store, rank=rank, world_size=world, timeout=to, **kw)
            else:
                dist.init_process_group(backend, init_method=f"tcp://{master_addr}:{master_port}", rank=rank,
                                        world_size=world, timeout=to, **kw)
        return cls(rank, world, device, backend=backend)

    @classmethod
    def from_env(cls, device: torch.device | str, timeout_s: float = 300.0) -> "Comm":
        """Rendezvous from the RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT that ``torchrun`` exports."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        return cls.init(rank, world, device, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                        int(os.environ.get("MASTER_PORT", "29500")), timeout_s)

    def subgroup(self, ranks: Sequence[int]) -> "Comm":
        """A communicator over ``ranks`` (collective call on the parent group)."""
        g = dist.new_group(list(ranks))
        r = list(ranks).index(self.rank) if self.rank in ranks else -1
        c = Comm(r, len(ranks), self.device, g, backend=self.backend)
        c._global_ranks = list(ranks)
        return c

    def _g(self, r: int) -> int:
        """Group rank -> global rank (P2P calls take global ranks)."""
        gr = getattr(self, "_global_ranks", None)
        return gr[r] if gr is not None else r

    def shutdown(self) -> None:
        if self.world > 1 and self.group is None and dist.is_initialized():
            dist.destroy_process_group()

    # ------------------------------------------------------------ collectives
    def _staged(self, t: torch.Tensor) -> bool:
        """gloo with device tensors (several ranks sharing one GPU, a test configuration): every transfer goes
        through host memory here rather than through gloo's own device path, which was seen to stall."""
        return self.composed and t.is_cuda

    def all_reduce(self, t: torch.Tensor, op=dist.ReduceOp.SUM, async_op: bool = False):
        if self.world == 1:
            return None
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out (1/world of inp) = sum over ranks of the matching slice of inp."""
        if self.world == 1:
            if out.data_ptr() != inp.data_ptr():
                out.copy_(inp)
            return None
        if self.composed:
            # gloo has no reduce_scatter: reduce the whole buffer, keep the local slice (asynchronous like the
            # native collective: the strategies post it from a backward hook and must not block backward there)
            n = out.numel()
            mine = inp[self.rank * n:(self.rank + 1) * n]
            w = self.all_reduce(inp, async_op=async_op)
            if out.data_ptr() == mine.data_ptr():
                return w
            if w is None:
                out.copy_(mine)
                return None
            return _StagedWork([w], [(out, mine)])
        return dist.reduce_scatter_tensor(out, inp, group=self.group, async_op=async_op)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        if self.world == 1:
            if out.data_ptr() != inp.data_ptr():
                out.copy_(inp)
            return None
        if self.composed:
            n = inp.numel()
            if inp.is_cuda:
                # gloo's device all-gather was seen to hang with several ranks on one GPU (both ranks inside the
                # call): stage through host memory, which is what gloo does internally for the other collectives
                src = inp.cpu()
                parts = [torch.empty_like(src) for _ in range(self.world)]
                dist.all_gather(parts, src, group=self.group)
                out[: n * self.world].copy_(torch.cat(parts))
                return None
            chunks = [out[i * n:(i + 1) * n] for i in range(self.world)]
            src = inp.clone() if inp.data_ptr() == chunks[self.rank].data_ptr() else inp
            return dist.all_gather(chunks, src, group=self.group, async_op=async_op)
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        if self.world == 1:
            out.copy_(inp)
            return None
        if self.composed:
            n = inp.numel() // self.world
            out[self.rank * n:(self.rank + 1) * n].copy_(inp[self.rank * n:(self.rank + 1) * n])
            peers = [r for r in range(self.world) if r != self.rank]
            for w in self.exchange([(inp[r * n:(r + 1) * n], r) for r in peers],
                                   [(out[r * n:(r + 1) * n], r) for r in peers]):
                w.wait()
            return None
        return dist.all_to_all_single(out, inp, group=self.group, async_op=async_op)

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        if self.world == 1:
            return None
        if self._staged(t):
            h = t.cpu()
            dist.broadcast(h, src=self._g(src), group=self.group)
            if self.rank != src:
                t.copy_(h)
            return None
        return dist.broadcast(t, src=self._g(src), group=self.group, async_op=async_op)

    def reduce(self, t: torch.Tensor, dst: int = 0, async_op: bool = False):
        if self.world == 1:
            return None
        if self._staged(t):
            h = t.cpu()
            dist.reduce(h, dst=self._g(dst), group=self.group)
            if self.rank == dst:
                t.copy_(h)
            return None
        return dist.reduce(t, dst=self._g(dst), group=self.group, async_op=async_op)

    def barrier(self) -> None:
        if self.world > 1:
            if self.device.type == "cuda" and not self.composed:
                dist.barrier(group=self.group, device_ids=[self.device.index or 0])
            else:
                dist.barrier(group=self.group)

    # -------------------------------------------------------------------- p2p
    def send(self, t: torch.Tensor, dst: int) -> None:
        dist.send(t.cpu() if self._staged(t) else t, self._g(dst), group=self.group)

    def recv(self, t: torch.Tensor, src: int) -> None:
        if self._staged(t):
            h = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(h, self._g(src), group=self.group)
            t.copy_(h)
            return
        dist.recv(t, self._g(src), group=self.group)

    def exchange(self, sends: Sequence[tuple[torch.Tensor, int]], recvs: Sequence[tuple[torch.Tensor, int]]):
        """Grouped point-to-point: post every send and receive as ONE batch (one RCCL group
        call, so transfers to different peers run concurrently over their own xGMI links)."""
        if not sends and not recvs:
            return []
        if self.composed and any(t.is_cuda for t, _ in list(sends) + list(recvs)):
            hosts = [(torch.empty(t.shape, dtype=t.dtype), s, t) for t, s in recvs]
            ops = [dist.P2POp(dist.isend, t.cpu(), self._g(d), group=self.group) for t, d in sends]
            ops += [dist.P2POp(dist.irecv, h, self._g(s), group=self.group) for h, s, _ in hosts]
            return [_StagedWork(dist.batch_isend_irecv(ops), [(t, h) for h, _, t in hosts])]
        ops = [dist.P2POp(dist.isend, t, self._g(d), group=self.group) for t, d in sends]
        ops += [dist.P2POp(dist.irecv, t, self._g(s), group=self.group) for t, s in recvs]
        return dist.batch_isend_irecv(ops)

    def all_reduce_scalar(self, v: float, op=dist.ReduceOp.SUM) -> float:
        if self.world == 1:
            return v
        host = self.device.type == "cpu" or self.composed
        t = torch.tensor([v], dtype=torch.float64 if host else torch.float32, device="cpu" if host else self.device)
        dist.all_reduce(t, op=op, group=self.group)
        return float(t)
