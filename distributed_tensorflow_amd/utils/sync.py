"""Thread synchronisation helpers for the parameter-server service threads."""
from __future__ import annotations

import threading
from typing import Callable


class BarrierBroken(RuntimeError):
    pass


class DynamicBarrier:
    """A reusable barrier whose number of parties can shrink while it is in use.

    The reference synchronises its per-worker server threads with a fixed
    ``threading.Barrier(n_nodes)`` (reference centralized/server.py:151); a worker that runs
    out of batches one step early, or dies, leaves the others waiting forever
    (server.py:95,113). Here a finished worker calls :meth:`leave`, a failed one
    :meth:`abort`, and every wait is bounded by ``timeout``.

    ``wait(action)``: the thread that completes the round runs ``action`` (while the others are
    still blocked) before anyone is released.
    """

    def __init__(self, parties: int, timeout: float | None = None):
        self._cond = threading.Condition()
        self.parties = parties
        self.timeout = timeout
        self._count = 0
        self._gen = 0
        self._broken = False
        self._action: Callable[[], None] | None = None

    def _trip(self) -> None:
        act, self._action = self._action, None
        try:
            if act is not None:
                act()
        except BaseException:
            self._broken = True
            self._cond.notify_all()
            raise
        self._count = 0
        self._gen += 1
        self._cond.notify_all()

    def wait(self, action: Callable[[], None] | None = None) -> None:
        with self._cond:
            if self._broken:
                raise BarrierBroken("barrier was aborted")
            gen = self._gen
            self._count += 1
            if action is not None:
                self._action = action
            if self._count >= self.parties:
                self._trip()
                return
            ok = self._cond.wait_for(lambda: self._gen != gen or self._broken, self.timeout)
            if not ok:
                self._broken = True
                self._cond.notify_all()
                raise BarrierBroken("barrier wait timed out")
            if self._broken and self._gen == gen:
                raise BarrierBroken("barrier was aborted")

    def leave(self) -> None:
        """The calling party will not take part in any further round."""
        with self._cond:
            self.parties -= 1
            if not self._broken and self.parties > 0 and self._count >= self.parties:
                self._trip()

    def abort(self) -> None:
        with self._cond:
            self._broken = True
            self._cond.notify_all()

    @property
    def broken(self) -> bool:
        return self._broken
