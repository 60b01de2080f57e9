"""Host-side (Python) profile of eager training steps of a small model: where the per-step CPU time goes.

usage: python tools/profile_host.py [resnet20|mlp] [steps]     (writes gpurun_out/host_profile_<model>.txt)
"""
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from distributed_tensorflow_amd.comm.backend import Comm
    from distributed_tensorflow_amd.engine.optimizer import SGD
    from distributed_tensorflow_amd.engine.replica import TrainConfig
    from distributed_tensorflow_amd.models.mlp import reference_mlp
    from distributed_tensorflow_amd.models.resnet import resnet20
    from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy

    name = sys.argv[1] if len(sys.argv) > 1 else "resnet20"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if name == "resnet20":
        model, x = resnet20(10), torch.randn(128, 32, 32, 3, device=dev).bfloat16()
        cfg = TrainConfig(param_dtype=torch.bfloat16, grad_dtype=torch.bfloat16)
    else:
        model, x = reference_mlp(dropout=0.0), torch.rand(100, 28, 28, 1, device=dev)
        cfg = TrainConfig()
    y = torch.randint(0, 10, (x.shape[0],), device=dev)
    st = MirroredStrategy(model, Comm(0, 1, dev), cfg, optimizer=SGD(0.05, momentum=0.9))
    for _ in range(10):
        st.train_step(x, y)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        st.train_step(x, y)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        st.train_step(x, y)
    torch.cuda.synchronize()
    pr.disable()
    out = io.StringIO()
    out.write(f"{name}: {wall:.3f} ms per eager step (unprofiled), {steps} profiled steps below\n")
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(35)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/host_profile_{name}.txt", "w") as f:
        f.write(out.getvalue())
    print(out.getvalue()[:6000])


if __name__ == "__main__":
    main()
