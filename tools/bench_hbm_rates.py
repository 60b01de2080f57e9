"""HBM rates of plain PyTorch kernels (fill, copy, read-modify, reduction) on a 1.64 GB bf16 tensor: the ceilings
that the framework's bandwidth-bound kernels are compared against.

    python tools/bench_hbm_rates.py
"""
import torch
d=torch.device('cuda:0')
def timed(fn,it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/it
n=3211264*256
y=torch.empty(n,device=d,dtype=torch.bfloat16); x=torch.randn(n//4,device=d).bfloat16()
x4=torch.empty(n,device=d,dtype=torch.bfloat16)
t=timed(lambda:y.zero_()); print("zero_ 1.64GB: %.1f us %.2f TB/s"%(t*1e3, n*2/t/1e9))
t=timed(lambda:y.fill_(1.0)); print("fill_ 1.64GB: %.1f us %.2f TB/s"%(t*1e3, n*2/t/1e9))
t=timed(lambda:y.copy_(x4)); print("copy 1.64GB->1.64GB: %.1f us %.2f TB/s total"%(t*1e3, 2*n*2/t/1e9))
yv=y.view(4,-1)
t=timed(lambda:torch.add(x, 1.0, out=yv[0])); print("read 0.41 write 0.41: %.1f us %.2f TB/s"%(t*1e3, 2*(n//4)*2/t/1e9))
t=timed(lambda:x4.sum()); print("read-only sum 1.64GB: %.1f us %.2f TB/s"%(t*1e3, n*2/t/1e9))
