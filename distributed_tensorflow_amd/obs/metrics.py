"""Structured run metrics.

The reference reports through ``print`` only (``elapsed time``, ``Test set
accuracy``, an iteration counter every 50 steps: reference
centralized/server.py:35,119, centralized/client.py:92-93,
decentralized/native/dist_keras.py:49-54) plus an optional supervisor socket
on port 4000 that nothing in the tree listens on (server.py:155-158,121-124).
:class:`MetricsLogger` writes one JSON object per line per rank (step, loss,
step latency, examples, bytes moved, staleness in async mode) and
:class:`SupervisorSink` replaces the supervisor hook with a working client for
the same framed-TCP messages.
"""
from __future__ import annotations

import json
import os
import socket
import struct
import time


class MetricsLogger:
    def __init__(self, path: str | None, rank: int = 0, flush_every: int = 20):
        self.rank = rank
        self.enabled = path is not None
        self._f = None
        self._n = 0
        self._flush_every = flush_every
        if path is not None:
            path = path.replace("%r", str(rank))
            d = os.path.dirname(path)
            if d:
                os.makedirs(d, exist_ok=True)
            self._f = open(path, "a", buffering=1 << 16)

    def log(self, **kw) -> None:
        if self._f is None:
            return
        kw.setdefault("ts", time.time())
        kw.setdefault("rank", self.rank)
        self._f.write(json.dumps(kw) + "\n")
        self._n += 1
        if self._n % self._flush_every == 0:
            self._f.flush()

    def summary(self, **kw) -> None:
        self.log(event="summary", **kw)
        if self._f is not None:
            self._f.flush()

    def close(self) -> None:
        if self._f is not None:
            self._f.close()
            self._f = None


class SupervisorSink:
    """Sends ``'start'``, ``('done', elapsed)`` and ``('results', accuracy)`` to a supervisor as
    4-byte big-endian length-prefixed JSON frames; connection failures are ignored (the
    supervisor is optional in the reference too)."""

    def __init__(self, address: str | None, port: int = 4000, timeout: float = 2.0):
        self.sock = None
        if address:
            try:
                self.sock = socket.create_connection((address, port), timeout=timeout)
            except OSError:
                self.sock = None

    def send(self, obj) -> None:
        if self.sock is None:
            return
        p = json.dumps(obj).encode()
        try:
            self.sock.sendall(struct.pack(">I", len(p)) + p)
        except OSError:
            self.sock = None

    def close(self) -> None:
        if self.sock is not None:
            self.sock.close()
            self.sock = None
