"""Shared pieces of ``bench.py`` and ``__graft_entry__.smoke``: the flagship training step.

Flagship = the headline configuration: ResNet-50 on synthetic 224x224 images, bf16,
decentralized ``keras``-style mirrored strategy (bucketed all-reduce on RCCL overlapped with
backward), SGD with momentum applied by the fused flat-arena kernel, fused softmax-cross-entropy
loss, fused batch-norm/ReLU/residual kernels and MFMA-GEMM 1x1 convolutions.
"""
from __future__ import annotations

import torch

from .comm.backend import Comm
from .engine.optimizer import SGD
from .engine.replica import TrainConfig
from .parallel.mirrored import MirroredStrategy


def make_resnet50_strategy(comm: Comm, classes: int = 1000, bucket_mb: int = 48, grad_dtype=torch.bfloat16,
                           lr: float = 0.1, seed: int = 0) -> MirroredStrategy:
    from .models.resnet import resnet50
    torch.manual_seed(seed)
    model = resnet50(classes)
    cfg = TrainConfig(loss_reduction="mean", param_dtype=torch.bfloat16, grad_dtype=grad_dtype,
                      bucket_bytes=bucket_mb << 20, first_bucket_bytes=8 << 20, seed=seed)
    opt = SGD(learning_rate=lr, momentum=0.9, weight_decay=1e-4, formula="torch")
    return MirroredStrategy(model, comm, cfg, optimizer=opt)


def synthetic_imagenet_batch(batch: int, device, size: int = 224, classes: int = 1000, seed: int = 0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch, size, size, 3, generator=g).to(device=device, dtype=torch.bfloat16)
    y = torch.randint(0, classes, (batch,), generator=g).to(device)
    return x, y


def smoke_step(device) -> float:
    """One tiny forward + backward + optimizer step of the flagship model."""
    comm = Comm(0, 1, device)
    st = make_resnet50_strategy(comm, classes=1000)
    x, y = synthetic_imagenet_batch(8, device, size=64)
    l0 = float(st.train_step(x, y))
    l1 = float(st.train_step(x, y))
    torch.cuda.synchronize(device)
    return l1 if l1 == l1 else l0
