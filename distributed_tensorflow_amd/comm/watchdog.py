"""Stuck-collective watchdog for the default communicator.

A synchronous data-parallel job that loses a peer does not fail by itself: the surviving ranks sit in a
collective (or, in the reference, in ``barrier.wait()`` -- reference centralized/server.py:95 -- forever).
``torch.distributed`` bounds each collective by the process-group timeout and, on the RCCL backend, polls the
communicator's asynchronous error state from its own thread; :func:`configure_async_errors` makes that path
abort the process instead of leaving it half alive. :class:`Watchdog` covers what is left -- a rank that stops
making progress for any reason (a hung kernel, a peer that stalls without dying, a gloo transfer that never
completes): the training loop calls :meth:`Watchdog.beat` once per step, and a host thread ends the process with
exit code 19 when no beat arrives within the limit, after writing every thread's stack to stderr. The launcher
sees the non-zero exit code, stops the other ranks and reports failure; nothing is re-executed.
"""
from __future__ import annotations

import faulthandler
import os
import sys
import threading
import time

EXIT_CODE = 19


def configure_async_errors() -> None:
    """RCCL asynchronous errors (``ncclCommGetAsyncError``, polled by the process group's own thread) and
    collective timeouts tear the process down rather than raising in a thread nobody joins."""
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    os.environ.setdefault("TORCH_NCCL_ENABLE_MONITORING", "1")


class Watchdog:
    def __init__(self, limit_s: float, rank: int = 0, poll=None, on_expire=None):
        """``poll``: optional callable returning an error string (or None), called every second -- the
        framework's own RCCL binding passes its ``ncclCommGetAsyncError`` query here.
        ``on_expire``: replaces the process exit (tests)."""
        self.limit_s = float(limit_s)
        self.rank = rank
        self.poll = poll
        self.on_expire = on_expire
        self._last = time.monotonic()
        self._what = "start"
        self._stop = threading.Event()
        self._paused = False
        self._thread = None
        if self.limit_s > 0:
            self._thread = threading.Thread(target=self._run, daemon=True, name="dtf-watchdog")
            self._thread.start()

    def beat(self, what: str = "") -> None:
        self._last = time.monotonic()
        self._what = what

    def pause(self) -> None:
        """Suspends the limit (evaluation, checkpoint writes: phases with no per-step beat)."""
        self._paused = True

    def resume(self) -> None:
        self._last = time.monotonic()
        self._paused = False

    def close(self) -> None:
        self._stop.set()

    def _expire(self, why: str) -> None:
        msg = f"[watchdog] rank {self.rank}: {why} (last progress: {self._what}); aborting\n"
        if self.on_expire is not None:
            self.on_expire(msg)
            return
        sys.stderr.write(msg)
        try:
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        except Exception:
            pass
        sys.stderr.flush()
        os._exit(EXIT_CODE)

    def _run(self) -> None:
        while not self._stop.wait(min(1.0, max(0.05, self.limit_s / 4))):
            if self.poll is not None:
                err = self.poll()
                if err:
                    self._expire(f"communicator error: {err}")
                    return
            if not self._paused and time.monotonic() - self._last > self.limit_s:
                self._expire(f"no training progress for {self.limit_s:.0f} s")
                return
