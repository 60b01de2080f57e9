"""Python face of the native framed-TCP transport (``csrc/runtime/transport.cpp``).

Plays the role of the reference's ``network.send_msg`` / ``recv_msg`` /
``recvall`` (reference centralized/network.py:4-28) for the control plane and
for the CPU plumbing mode: a message is ``(kind, meta dict, tensor payload)``;
payload bytes go straight from / into tensor memory, nothing is pickled.
"""
from __future__ import annotations

import ctypes
import json

import torch

from ..ops import _lib

OK, ERR_SYS, ERR_TIMEOUT, ERR_CLOSED, ERR_PROTOCOL = 0, -1, -2, -3, -4


class TransportError(RuntimeError):
    def __init__(self, code: int, what: str):
        names = {ERR_SYS: "system error", ERR_TIMEOUT: "timed out", ERR_CLOSED: "connection closed by peer",
                 ERR_PROTOCOL: "protocol error"}
        super().__init__(f"{what}: {names.get(code, code)}")
        self.code = code


class ConnectionClosed(TransportError):
    pass


def _rt():
    lib = _lib.runtime()
    if not getattr(lib, "_tcp_declared", False):
        lib.dtf_tcp_listen.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        lib.dtf_tcp_accept.argtypes = [ctypes.c_int, ctypes.c_double]
        lib.dtf_tcp_connect.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_double, ctypes.c_double]
        lib.dtf_tcp_send_msg.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                                         ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double]
        lib.dtf_tcp_recv_head.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32),
                                          ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64),
                                          ctypes.c_double]
        lib.dtf_tcp_recv_body.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                          ctypes.c_uint64, ctypes.c_double]
        lib.dtf_tcp_close.argtypes = [ctypes.c_int]
        lib.dtf_tcp_bound_port.argtypes = [ctypes.c_int]
        lib._tcp_declared = True
    return lib


def _raise(code: int, what: str):
    if code == ERR_CLOSED:
        raise ConnectionClosed(code, what)
    raise TransportError(code, what)


class Connection:
    def __init__(self, fd: int, timeout: float = -1.0):
        self.fd = fd
        self.timeout = timeout

    def send(self, kind: int, meta: dict | None = None, payload: torch.Tensor | None = None) -> None:
        mb = json.dumps(meta or {}).encode()
        if payload is not None:
            if payload.is_cuda:
                payload = payload.cpu()
            payload = payload.contiguous()
            nbytes = payload.numel() * payload.element_size()
            pp = payload.data_ptr()
        else:
            nbytes, pp = 0, 0
        rc = _rt().dtf_tcp_send_msg(self.fd, kind, mb, len(mb), pp, nbytes, self.timeout)
        if rc != 0:
            _raise(rc, "send")

    def recv(self, into: torch.Tensor | None = None, timeout: float | None = None):
        """Returns ``(kind, meta, payload)``. ``into`` is a preallocated CPU tensor the payload is
        received into (its dtype decides how the bytes are viewed); without it a uint8 tensor
        of the right size is allocated."""
        t = self.timeout if timeout is None else timeout
        kind, ml, pl = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        rc = _rt().dtf_tcp_recv_head(self.fd, ctypes.byref(kind), ctypes.byref(ml), ctypes.byref(pl), t)
        if rc != 0:
            _raise(rc, "recv")
        mbuf = ctypes.create_string_buffer(ml.value) if ml.value else None
        n = pl.value
        payload = None
        if n:
            if into is not None:
                cap = into.numel() * into.element_size()
                if cap < n or into.is_cuda or not into.is_contiguous():
                    raise ValueError(f"receive buffer of {cap} bytes cannot take a {n}-byte payload")
                payload = into.view(-1)[: n // into.element_size()]
                pp = into.data_ptr()
            else:
                payload = torch.empty(n, dtype=torch.uint8)
                pp = payload.data_ptr()
        else:
            pp = 0
        rc = _rt().dtf_tcp_recv_body(self.fd, mbuf, ml.value, pp, n, t)
        if rc != 0:
            _raise(rc, "recv body")
        meta = json.loads(mbuf.raw.decode()) if mbuf is not None else {}
        return kind.value, meta, payload

    def close(self) -> None:
        if self.fd >= 0:
            _rt().dtf_tcp_close(self.fd)
            self.fd = -1


class Listener:
    def __init__(self, address: str = "127.0.0.1", port: int = 6000, backlog: int = 50):
        self.fd = _rt().dtf_tcp_listen(address.encode(), port, backlog)
        if self.fd < 0:
            raise TransportError(self.fd, f"listen on {address}:{port}")
        self.port = _rt().dtf_tcp_bound_port(self.fd)

    def accept(self, timeout: float = -1.0, conn_timeout: float = -1.0) -> Connection:
        fd = _rt().dtf_tcp_accept(self.fd, timeout)
        if fd < 0:
            _raise(fd, "accept")
        return Connection(fd, conn_timeout)

    def close(self) -> None:
        if self.fd >= 0:
            _rt().dtf_tcp_close(self.fd)
            self.fd = -1


def connect(address: str, port: int, timeout: float = 60.0, retry_interval: float = 0.2,
            conn_timeout: float = -1.0) -> Connection:
    """Connects, retrying while the server is not listening yet (the reference worker's
    1-second ``ConnectionRefusedError`` loop, reference centralized/client.py:55-62)."""
    fd = _rt().dtf_tcp_connect(address.encode(), port, timeout, retry_interval)
    if fd < 0:
        _raise(fd, f"connect to {address}:{port}")
    return Connection(fd, conn_timeout)
