"""Two-norm backward apply kernel with one and with two 16-byte items per thread, on the first down-sampling block
of ResNet-50 (256 channels at 56 x 56): run under `rocprofv3 --kernel-trace --stats` and compare the rows
bn_apply_bwd2_kernel<1, ..> and <2, ..>.   usage: python tools/micro/bn_dual_rows.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from distributed_tensorflow_amd.models import resnet as R  # noqa: E402
from distributed_tensorflow_amd.ops import _lib  # noqa: E402

dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
net = torch.nn.Sequential(R.Bottleneck(64, 64, 1), R.Bottleneck(256, 64, 1)).to(dev)
for m in net.modules():
    if isinstance(m, R.Conv):
        m.weight.data = m.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
net.train()
x = torch.randn(batch, 64, 56, 56, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
for rows in (2, 1, 2, 1):
    _lib.kernels().dtf_set_bn_apply2_rows(rows)
    for _ in range(4):
        for p in net.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        y = net(xi)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
print("done")
