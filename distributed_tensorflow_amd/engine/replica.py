"""One model replica on one device: arenas, loss, forward/backward.

Every strategy (parameter server or decentralized) drives the same object:

    rep = Replica(model, device, cfg)
    loss = rep.forward_backward(x, y)      # gradients land in rep.flat.grad, bucket by bucket
    ... strategy-specific communication + optimizer ...

Loss semantics follow the reference: the loss is the per-example sparse
categorical cross-entropy vector (reference centralized/client.py:11-13) and
its gradient is taken of the SUM over the batch in the parameter-server path
(client.py:16-19: ``tape.gradient`` of a vector target) or of the MEAN in the
Keras ``fit`` path (decentralized/native/dist_keras.py:30-32,42).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable

import torch

from ..api.model import Model, as_model
from ..ops.softmax_ce import softmax_ce_eval, softmax_cross_entropy
from .flat import BucketPlanner, FlatParams
from .reducer import GradReducer
from ..obs.trace import phase


@dataclass
class TrainConfig:
    loss: str | Callable = "sparse_categorical_crossentropy"
    loss_reduction: str = "mean"          # 'mean' | 'sum'
    training: bool = True                 # run layers (dropout, batch norm) in training mode
    probs_clip_eps: float | None = None   # reference-exact loss on clipped probabilities
    param_dtype: torch.dtype | None = None  # cast floating parameters (e.g. torch.bfloat16); None keeps them
    keep_fp32: tuple = ("bn", "norm", "bias_fp32")  # name fragments of parameters that stay fp32 under a bf16 cast
    grad_dtype: torch.dtype = torch.float32
    bucket_bytes: int = 48 << 20
    first_bucket_bytes: int | None = 8 << 20
    last_bucket_bytes: int | None = 4 << 20   # the last bucket's collective is exposed: keep it small
    shard_ways: int = 1
    split_dtypes: bool = False            # arena layout: bf16 parameters first, fp32 after them (sharded parameter server)
    seed: int = 0
    graph: bool = False                   # capture the whole train step into a hipGraph (GPU, static shapes)
    verify_replicas: int = 0              # debug: compare the master-weight checksum across ranks every N steps
    extra: dict = field(default_factory=dict)


def _cast_params(model: torch.nn.Module, dtype: torch.dtype, keep_fp32: tuple) -> None:
    norm_types = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.LayerNorm)
    for mod in model.modules():
        if isinstance(mod, norm_types) or getattr(mod, "keep_fp32", False):
            continue
        for n, p in mod._parameters.items():
            if p is not None and p.is_floating_point() and not any(k in n for k in keep_fp32):
                p.data = p.data.to(dtype)


class Replica:
    def __init__(self, model, device, cfg: TrainConfig | None = None):
        self.cfg = cfg or TrainConfig()
        self.device = torch.device(device)
        self.model: Model = as_model(model).to(self.device)
        if self.cfg.param_dtype is not None and self.cfg.param_dtype != torch.float32:
            _cast_params(self.model, self.cfg.param_dtype, self.cfg.keep_fp32)
        self.flat = FlatParams.from_module(self.model, grad_dtype=self.cfg.grad_dtype, shard_ways=self.cfg.shard_ways,
                                           split_dtypes=self.cfg.split_dtypes)
        self.planner = BucketPlanner(self.flat, self.cfg.bucket_bytes, self.cfg.first_bucket_bytes,
                                     self.cfg.last_bucket_bytes)
        self.reducer = GradReducer(self.flat, self.planner)
        self.reducer.attach()
        self.step_count = 0

    # ------------------------------------------------------------------ loss
    def _loss_vector(self, x, y) -> torch.Tensor:
        cfg = self.cfg
        if callable(cfg.loss):
            return cfg.loss(self.model, x, y)
        if cfg.loss not in ("sparse_categorical_crossentropy", "sparse_ce", "mlm"):
            raise ValueError(f"unknown loss {cfg.loss!r}")
        if self.model.supports_logits:
            logits = self.model(x, logits=True)
            return softmax_cross_entropy(logits, y, reduction="none", probs_clip_eps=cfg.probs_clip_eps)
        probs = self.model(x).float()
        eps = cfg.probs_clip_eps if cfg.probs_clip_eps is not None else 1e-7
        p = probs.reshape(-1, probs.shape[-1]).clamp(eps, 1.0 - eps)
        p = p / p.sum(dim=1, keepdim=True)
        return -torch.log(p.gather(1, y.reshape(-1, 1).long())[:, 0])

    def forward_backward(self, x, y, on_bucket_ready=None, grad_scale: float = 1.0) -> torch.Tensor:
        """Forward + backward; returns the detached scalar loss (mean over the batch, for logging).
        ``grad_scale`` is folded into the bucket packing kernel (e.g. 1/world)."""
        if self.model.training != self.cfg.training:   # walks every module: only when the mode actually changes
            self.model.train(self.cfg.training)
        self.reducer.on_ready = on_bucket_ready
        self.reducer.grad_scale = grad_scale
        self.reducer.begin_step()
        with phase("forward"):
            lv = self._loss_vector(x, y)
        if self.cfg.loss_reduction == "sum":
            total = lv.sum()
            shown = total.detach() / max(1, lv.numel())
        else:
            if self.cfg.loss == "mlm":  # rows with label < 0 are ignored and do not count
                denom = (y >= 0).sum().clamp(min=1)
                total = lv.sum() / denom
            else:
                total = lv.mean()
            shown = total.detach()
        with phase("backward"):
            total.backward()
        self.step_count += 1
        return shown

    def finish_backward(self):
        return self.reducer.finish()

    # ------------------------------------------------------------------ eval
    @torch.no_grad()
    def evaluate(self, dataset, max_batches: int | None = None) -> tuple[float, float, int]:
        """(mean loss, accuracy, examples) over ``dataset`` -- the argmax-accuracy loop of the
        reference's ``test`` helper (reference centralized/server.py:24-37)."""
        self.model.eval()
        tot_loss = torch.zeros((), dtype=torch.float64, device=self.device)
        correct = torch.zeros((), dtype=torch.int64, device=self.device)
        n = 0
        it = dataset.iterate(self.device) if hasattr(dataset, "iterate") else iter(dataset)
        for i, batch in enumerate(it):
            if max_batches is not None and i >= max_batches:
                break
            x, y = (batch[0] if len(batch) == 2 else tuple(batch[:-1])), batch[-1]
            x = tuple(t.to(self.device) for t in x) if isinstance(x, tuple) else x.to(self.device)
            y = y.to(self.device)
            if self.model.supports_logits:
                out = self.model(x, logits=True)
                l, c = softmax_ce_eval(out, y if y.dtype in (torch.int32, torch.int64) else y.long())
                tot_loss += l.double().sum()
                correct += c.long().sum()
            else:
                p = self.model(x).float()
                correct += (p.argmax(-1).reshape(-1) == y.reshape(-1)).sum()
                tot_loss += -torch.log(p.reshape(-1, p.shape[-1]).gather(1, y.reshape(-1, 1).long()).clamp_min(1e-7)).double().sum()
            n += y.numel()
        n = max(n, 1)
        return float(tot_loss) / n, float(correct) / n, n
