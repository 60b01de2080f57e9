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
                       grad_dtype=torch.bfloat16, seed: int = 0, strategy: str = "custom:direct"):
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


class HostInputPipeline:
    """Synthetic ImageNet-shaped input that really travels: uint8 images live in HOST memory, the native loader's
    threads (``csrc/runtime/loader.cpp``) gather each batch by random index into pinned staging, a copy stream uploads
    it one step ahead of its use and casts / normalises it to bf16 on the GPU. Two device buffer sets alternate; the
    upload into a set waits (on the GPU, by event) for the step that last read it.

    ``next()`` returns the batch for the coming step. The reference feeds every step from host memory as well
    (``tf.data`` batches of MNIST, reference initializer.py:25-49)."""

    def __init__(self, batch: int, device, size: int = 224, classes: int = 1000, pool: int | None = None, seed: int = 0,
                 threads: int = 4):
        import numpy as np

        from .data.loader import NativeLoader
        self.batch, self.device = batch, torch.device(device)
        pool = pool or max(2 * batch, 2048)
        rng = np.random.default_rng(seed)
        self._x = rng.integers(0, 256, (pool, size, size, 3), dtype=np.uint8)
        self._y = rng.integers(0, classes, (pool,), dtype=np.int64)
        self._rng, self._pool = rng, pool
        self.loader = NativeLoader(self._x, self._y, batch, to_float=False, slots=4, threads=threads)
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(self.device) if self.cuda else None
        dt = torch.bfloat16
        self._raw = [torch.empty((batch, size, size, 3), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._img = [torch.empty((batch, size, size, 3), dtype=dt, device=self.device) for _ in range(2)]
        self._lab = [torch.empty((batch,), dtype=torch.int64, device=self.device) for _ in range(2)]
        self._read_done = [None, None]       # main-stream event: the step that read set k has been enqueued completely
        self._uploaded = [None, None]        # (copy-stream event, loader slot) of the batch in set k
        self._slots: list[int] = []
        self._i = 0
        for _ in range(2):
            self._submit()
        self._upload(0)

    def _submit(self) -> None:
        self._slots.append(self.loader.submit(self._rng.integers(0, self._pool, self.batch)))

    def _upload(self, k: int) -> None:
        slot = self._slots.pop(0)
        hx, hy = self.loader.take(slot)      # blocks until the loader threads have assembled the batch
        if not self.cuda:
            self._raw[k].copy_(hx)
            self._lab[k].copy_(hy)
            self._img[k].copy_((self._raw[k].float() - 127.5) / 64.0)
            self.loader.release(slot)
            self._uploaded[k] = (None, None)
            return
        with torch.cuda.stream(self.side):
            if self._read_done[k] is not None:
                self.side.wait_event(self._read_done[k])
            self._raw[k].copy_(hx, non_blocking=True)
            self._lab[k].copy_(hy, non_blocking=True)
            self._img[k].copy_(self._raw[k])             # uint8 -> bf16 (0..255 are exact in bf16)
            self._img[k].sub_(127.5).mul_(1.0 / 64.0)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self._uploaded[k] = (ev, slot)

    def next(self):
        k = self._i & 1
        if self.cuda and self._i > 0:        # everything enqueued so far belongs to steps that read the other set
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._read_done[k ^ 1] = ev
        self._submit()
        ev, slot = self._uploaded[k]
        self._upload(k ^ 1)                  # the batch after this one, while this step computes
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            ev.synchronize()                 # issued a whole step ago: the pinned staging slot is free again
            self.loader.release(slot)
        self._i += 1
        return self._img[k], self._lab[k]

    def close(self) -> None:
        if self.cuda:
            torch.cuda.synchronize(self.device)
        self.loader.close()


def smoke_step(device) -> float:
    """One tiny forward + backward + optimizer step of the flagship model."""
    comm = Comm(0, 1, device)
    st = make_resnet50_strategy(comm, classes=1000)
    x, y = synthetic_imagenet_batch(8, device, size=64)
    l0 = float(st.train_step(x, y))
    l1 = float(st.train_step(x, y))
    torch.cuda.synchronize(device)
    return l1 if l1 == l1 else l0
