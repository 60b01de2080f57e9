"""Step-phase tracing: roctx ranges for rocprofv3 plus device-event phase timers.

``with phase("backward"):`` pushes a roctx range (visible in ``rocprofv3 --marker-trace`` next to
the kernel trace) when the roctx library can be loaded, and -- when a :class:`PhaseTimer` is
active -- brackets the phase with device events so the per-phase device time ends up in the
metrics log. The reference's only timing is two ``time.time()`` brackets around the whole run
(reference centralized/server.py:77-78,117, decentralized/native/dist_keras.py:41-43).

Recipes (run on the GPU box, from /tmp with TMPDIR=/tmp):
  per-kernel time     rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 bench.py --steps 4
  MFMA / LDS counters rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d OUT -- python3 tools/bench_kernels.py --quick
  HBM traffic         rocprofv3 --pmc FETCH_SIZE --output-format csv ...   (FETCH_SIZE reads half the bytes of a wide
                      coalesced stream on gfx950: double it) and, in a separate run, --pmc WRITE_SIZE
``tools/summarize_trace.py`` turns a kernel trace into a per-step breakdown.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch

_roctx = None
_roctx_tried = False
_active_timer = None


def _load_roctx():
    global _roctx, _roctx_tried
    if _roctx_tried:
        return _roctx
    _roctx_tried = True
    if os.environ.get("DTF_ROCTX", "1") != "1":
        return None
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            _roctx = lib
            break
        except OSError:
            continue
    return _roctx


class PhaseTimer:
    """Accumulates device time per phase name with CUDA/HIP events (read with :meth:`report`)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.pending: list[tuple[str, object, object]] = []
        self.totals: dict[str, float] = {}
        self.counts: dict[str, int] = {}

    def __enter__(self):
        global _active_timer
        _active_timer = self
        return self

    def __exit__(self, *a):
        global _active_timer
        _active_timer = None

    def report(self) -> dict[str, float]:
        """Milliseconds PER STEP of every phase. A step is counted by its ``forward`` phase; a phase that occurs
        several times in a step (``comm_in_hook``: once per gradient bucket) is the sum of its occurrences. Without
        a ``forward`` phase (a loop that is not a training step) every phase is averaged over its own occurrences."""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        for name, s, e in self.pending:
            ms = s.elapsed_time(e) if self.device.type == "cuda" else (e - s) * 1e3
            self.totals[name] = self.totals.get(name, 0.0) + ms
            self.counts[name] = self.counts.get(name, 0) + 1
        self.pending.clear()
        steps = self.counts.get("forward", 0)
        return {k: v / (steps if steps else self.counts[k]) for k, v in self.totals.items()}


@contextlib.contextmanager
def phase(name: str):
    lib = _load_roctx()
    if lib is not None:
        lib.roctxRangePushA(name.encode())
    t = _active_timer
    if t is not None:
        if t.device.type == "cuda":
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
        else:
            import time
            s = time.perf_counter()
    try:
        yield
    finally:
        if t is not None:
            if t.device.type == "cuda":
                e.record()
            else:
                import time
                e = time.perf_counter()
            t.pending.append((name, s, e))
        if lib is not None:
            lib.roctxRangePop()
