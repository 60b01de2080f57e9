"""CPU-affinity pinning: the ``-ca`` flag.

The reference pins the parameter server to core 0, worker ``index`` to core ``index + 1``
and a mirrored worker to core ``index`` with ``psutil.Process().cpu_affinity([k])``
(reference centralized/server.py:144-146, centralized/client.py:48-50,
decentralized/native/dist_keras.py:63-65) to imitate separate machines on one host.
"""
from __future__ import annotations

import os


def pin_to_core(core: int) -> int | None:
    """Restricts the calling process to one core (modulo the cores it may use); also caps the
    intra-op thread pools at one thread. Returns the core chosen, or None if unsupported."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover - non-Linux
        return None
    if not allowed:
        return None
    chosen = allowed[core % len(allowed)]
    os.sched_setaffinity(0, {chosen})
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        import torch
        torch.set_num_threads(1)
    except Exception:  # pragma: no cover
        pass
    return chosen
