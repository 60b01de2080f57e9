"""Data gradient of the stride-2 3x3 convolutions of ResNet-50 (parity-class row order, k-tiles of unreachable taps
skipped) timed on its own, with the kernel's measurement hooks: lean / general store phase, LDS-DMA main loop.
usage: python tools/micro/dgrad_stride2.py [batch] [quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("DTF_WGRAD_STREAM", "0")
from distributed_tensorflow_amd.ops import _lib, conv  # noqa: E402
from distributed_tensorflow_amd.ops import gemm as G  # noqa: E402

dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = _lib.kernels()
KIND = {0: "gemm", 1: "fwd", 2: "dgrad", 3: "wgrad", 4: "wgradT"}


def run(c, hw, label, reps=5):
    torch.manual_seed(0)
    x = torch.randn(batch, c, hw, hw, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(c, c, 3, 3, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    times = {}
    for it in range(reps + 2):
        G.launch_log = [] if it >= 2 else None
        xi = x.clone().requires_grad_(True)
        wi = w.clone().requires_grad_(True)
        y = conv.conv2d(xi, wi, 2, 1)
        y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        if it >= 2:
            for d, e0, e1 in G.launch_log:
                times.setdefault(KIND[d["conv"]], []).append(e0.elapsed_time(e1) * 1e3)
    G.launch_log = None
    flop = 2.0 * batch * (hw // 2) ** 2 * c * c * 9
    out = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    print(f"{label:28s} {c:4d} ch @{hw}: " + "  ".join(f"{k} {t:7.1f} us {flop / t / 1e6:5.0f} TF" for k, t in out.items()),
          flush=True)
    return xi.grad


quick = len(sys.argv) > 2 and sys.argv[2] == "quick"      # the default kernels only (counter runs)
for c, hw in ((128, 56), (256, 28), (512, 14)):
    ref = run(c, hw, "default")
    if quick:
        continue
    lib.dtf_gemm_set_lean(0)
    g = run(c, hw, "general store phase")
    lib.dtf_gemm_set_lean(1)
    assert torch.equal(g, ref)
    lib.dtf_gemm_set_stages(3)
    g = run(c, hw, "LDS-DMA main loop")
    lib.dtf_gemm_set_stages(0)
    assert torch.equal(g, ref)
