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
    # backward runs on the autograd engine's own thread, which cProfile does not see: time every custom Function's
    # backward with a wrapper instead (host time only; kernels are asynchronous)
    import collections
    from distributed_tensorflow_amd.ops import conv as _c, norm as _n, dense as _d, pool as _p, softmax_ce as _s
    acc = collections.defaultdict(lambda: [0, 0.0])
    inner = cProfile.Profile() if os.environ.get("PROFILE_BACKWARD") == "1" else None   # function-level view of backward
    patched = []
    for mod in (_c, _n, _d, _p, _s):
        for nm in dir(mod):
            cls = getattr(mod, nm)
            if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
                orig = cls.backward

                def wrap(orig=orig, key=f"{mod.__name__.rsplit('.', 1)[-1]}.{nm}"):
                    def timed(*a, **k):
                        t = time.perf_counter()
                        if inner is not None:
                            inner.enable()
                            r = orig(*a, **k)
                            inner.disable()
                        else:
                            r = orig(*a, **k)
                        e = acc[key]
                        e[0] += 1
                        e[1] += time.perf_counter() - t
                        return r
                    return staticmethod(timed)
                cls.backward = wrap()
                patched.append((cls, orig))
    t0 = time.perf_counter()
    for _ in range(steps):
        st.train_step(x, y)
    torch.cuda.synchronize()
    wall_b = (time.perf_counter() - t0) / steps * 1e3
    for cls, orig in patched:
        cls.backward = staticmethod(orig)
    bw_lines = [f"backward host time per step (wrapped run: {wall_b:.3f} ms per step)\n"]
    for key, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        bw_lines.append(f"  {key:40s} {n / steps:6.1f} calls  {t / steps * 1e3:7.3f} ms  ({t / max(n, 1) * 1e6:5.1f} us each)\n")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        st.train_step(x, y)
    torch.cuda.synchronize()
    pr.disable()
    out = io.StringIO()
    out.write(f"{name}: {wall:.3f} ms per eager step (unprofiled), {steps} profiled steps below\n")
    out.write("".join(bw_lines))
    if inner is not None:
        out.write("inside the backward functions (profiled on the autograd thread):\n")
        pstats.Stats(inner, stream=out).sort_stats("tottime").print_stats(30)
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(35)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/host_profile_{name}.txt", "w") as f:
        f.write(out.getvalue())
    print(out.getvalue()[:6000])


if __name__ == "__main__":
    main()
