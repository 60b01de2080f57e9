"""Shared pieces of ``bench.py`` and ``__graft_entry__.smoke``: the benchmark training steps.

Flagship = the headline configuration: ResNet-50 on synthetic 224x224 images, bf16,
decentralized ``keras``-style mirrored strategy (bucketed all-reduce on RCCL overlapped with
backward), SGD with momentum applied by the fused flat-arena kernel, fused softmax-cross-entropy
loss, fused batch-norm/ReLU/residual kernels and MFMA implicit-GEMM convolutions. ``make_strategy``
builds the same models under every other strategy of the framework (custom all-reduce algorithms, the
peer-pointer kernels, graph gossip, synchronous and asynchronous parameter server), and BERT-base MLM
(sequence length 512) for the "custom bucketed all-reduce overlapped with backward" configuration.
"""
from __future__ import annotations

import torch

from .comm.backend import Comm
from .engine.optimizer import SGD, Adam
from .engine.replica import TrainConfig
from .parallel.mirrored import CustomAllReduceStrategy, GraphStrategy, MirroredStrategy

BUCKET_MB = 48          # per-link transfers of a bucket stay multi-MiB on a fully connected xGMI node


def make_strategy(kind: str, model, comm: Comm, cfg: TrainConfig, optimizer):
    """``kind``: ``keras`` | ``custom`` / ``custom:<algorithm>`` (direct, ring, multiring, tree, oneshot on grouped
    send/recv) | ``ipc`` / ``ipc-<schedule>`` (kernels on HIP-IPC peer pointers) | ``graph`` / ``graph:<degree>`` |
    ``sync`` | ``async``."""
    from .parallel import ps
    name, _, arg = kind.partition(":")
    if name == "keras":
        return MirroredStrategy(model, comm, cfg, optimizer=optimizer)
    if name == "custom":
        return CustomAllReduceStrategy(model, comm, cfg, optimizer=optimizer, algorithm=arg or "auto")
    if name.startswith("ipc"):
        return CustomAllReduceStrategy(model, comm, cfg, optimizer=optimizer, algorithm=name)
    if name == "graph":
        return GraphStrategy(model, comm, cfg, optimizer=optimizer, degree=int(arg) if arg else 2)
    if name == "sync":
        return ps.SyncPSStrategy(model, comm, cfg, optimizer=optimizer)
    if name == "async":
        return ps.AsyncPSStrategy(model, comm, cfg, optimizer=optimizer)
    raise ValueError(f"unknown strategy {kind!r}")


def make_resnet50_strategy(comm: Comm, classes: int = 1000, bucket_mb: int = BUCKET_MB, grad_dtype=torch.bfloat16,
                           lr: float = 0.1, seed: int = 0, strategy: str = "keras"):
    from .models.resnet import resnet50
    torch.manual_seed(seed)
    model = resnet50(classes)
    cfg = TrainConfig(loss_reduction="mean", param_dtype=torch.bfloat16, grad_dtype=grad_dtype,
                      bucket_bytes=bucket_mb << 20, first_bucket_bytes=8 << 20, seed=seed)
    opt = SGD(learning_rate=lr, momentum=0.9, weight_decay=1e-4, formula="torch")
    return make_strategy(strategy, model, comm, cfg, opt)


def make_bert_strategy(comm: Comm, size: str = "base", seq_len: int = 512, bucket_mb: int = BUCKET_MB,
                       grad_dtype=torch.bfloat16, seed: int = 0, strategy: str = "custom:ring"):
    from .models import bert
    torch.manual_seed(seed)
    model = bert.bert_base(seq_len) if size == "base" else bert.bert_tiny(seq_len)
    cfg = TrainConfig(loss="mlm", loss_reduction="mean", param_dtype=torch.bfloat16, grad_dtype=grad_dtype,
                      bucket_bytes=bucket_mb << 20, first_bucket_bytes=8 << 20, seed=seed)
    return make_strategy(strategy, model, comm, cfg, Adam(learning_rate=1e-4))


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
