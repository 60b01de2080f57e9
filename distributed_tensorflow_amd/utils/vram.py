"""First touch of the device memory, in a throwaway process.

Measured on MI355X boxes fresh from boot: the FIRST process that works through tens of GiB of HBM runs the
two-stream ResNet-50 step 1.0-1.5 % slower than every process after it, for as long as it lives (56 s of warm-up
steps do not help, nor does building, stepping, freeing and rebuilding the whole model inside the process); a
process that merely allocated and zeroed the memory once and exited is enough to make the next one run at full
speed. ``pretouch`` is that process: it starts a child that allocates most of the free memory of one device,
writes it once and exits. It must run before the caller initialises the GPU itself (the child needs the memory),
costs a few seconds, and failing is harmless (the caller just runs as it would have).
"""
from __future__ import annotations

import os
import subprocess
import sys
import time

_CHILD = r"""
import sys, torch
dev = int(sys.argv[1]); frac = float(sys.argv[2])
torch.cuda.set_device(dev)
free, total = torch.cuda.mem_get_info(dev)
n = int(free * frac) // (1 << 20) * (1 << 20)
done = 0
# a few large pieces rather than one: an allocation that fails only loses its own piece
piece = max(1 << 30, n // 8)
held = []
while done + piece <= n:
    try:
        t = torch.empty(piece, dtype=torch.uint8, device="cuda")
    except RuntimeError:
        break
    t.zero_()
    held.append(t)
    done += piece
torch.cuda.synchronize(dev)
print("%.1f" % (done / (1 << 30)))
"""


def pretouch(device_index: int = 0, fraction: float = 0.9, timeout_s: float = 120.0) -> dict:
    """Runs the child for device ``device_index``; returns ``{"gib": touched GiB (0.0 on failure), "seconds": wall
    time, "ok": bool}`` (and ``"error"``: the child's last line of stderr when it failed). Never raises."""
    t0 = time.perf_counter()
    out = {"gib": 0.0, "seconds": 0.0, "ok": False}
    try:
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE",
                  "TORCHELASTIC_RUN_ID"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, "-c", _CHILD, str(int(device_index)), str(float(fraction))], env=env,
                           capture_output=True, text=True, timeout=timeout_s)
        if r.returncode == 0:
            out["gib"] = float(r.stdout.strip().splitlines()[-1])
            out["ok"] = out["gib"] > 0
        else:
            out["error"] = (r.stderr.strip().splitlines() or ["exit code %d" % r.returncode])[-1][:200]
    except Exception as e:      # a missing GPU, a timeout, a busy device: the caller carries on without
        out["error"] = repr(e)[:200]
    out["seconds"] = round(time.perf_counter() - t0, 2)
    return out
