"""Optimizers over flat arenas.

An optimizer *spec* (:class:`Adam`, :class:`SGD`) is a plain description, like the
``tf.keras.optimizers.Adam()`` object the reference passes around (reference
centralized/server.py:52-55, decentralized/native/dist_keras.py:30).
:class:`FlatOptimizer` binds a spec to a :class:`FlatParams` and owns the
moment arenas; ``step()`` is one fused kernel launch per contiguous range.

A parameter-server shard owner calls ``step(ranges=[its shard of every
bucket])``; the mirrored strategy calls ``step()`` over everything. The
reference-exact sync-PS mode needs one optimizer *state* per worker applied to
shared variables (server.py:52-55,170): create several ``FlatOptimizer`` on
the same ``FlatParams``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence

import torch

from ..ops import optim as K
from .flat import FlatParams


@dataclass
class Adam:
    """Keras defaults (learning_rate 1e-3, beta_1 0.9, beta_2 0.999, epsilon 1e-7)."""
    learning_rate: float = 1e-3
    beta_1: float = 0.9
    beta_2: float = 0.999
    epsilon: float = 1e-7
    weight_decay: float = 0.0
    decoupled_weight_decay: bool = False
    formula: str = "keras"  # 'keras' | 'torch'
    kind: str = "adam"


@dataclass
class SGD:
    learning_rate: float = 0.01
    momentum: float = 0.0
    nesterov: bool = False
    weight_decay: float = 0.0
    formula: str = "keras"  # 'keras' | 'torch'
    kind: str = "sgd"


def resolve_spec(opt) -> Adam | SGD:
    if opt is None:
        return Adam()
    if isinstance(opt, (Adam, SGD)):
        return opt
    if isinstance(opt, str):
        name = opt.lower()
        if name == "adam":
            return Adam()
        if name in ("sgd", "momentum"):
            return SGD(momentum=0.9 if name == "momentum" else 0.0)
    raise ValueError(f"unknown optimizer {opt!r}; use 'adam', 'sgd' or an Adam/SGD spec")


class FlatOptimizer:
    def __init__(self, flat: FlatParams, spec=None, grad_scale: float = 1.0,
                 state_ranges: Sequence[tuple[int, int]] | None = None):
        """``state_ranges``: the element ranges this instance will ever update (a PS shard owner
        passes its shards); moments are allocated for the whole index space only when omitted."""
        self.flat = flat
        self.spec = resolve_spec(spec)
        s = self.spec
        dev = flat.device
        if s.kind == "adam":
            self.hyper = K.OptHyper(dev, lr=s.learning_rate, beta1=s.beta_1, beta2=s.beta_2, eps=s.epsilon,
                                    weight_decay=s.weight_decay, grad_scale=grad_scale)
            self.mode = (K.ADAM_TORCH if s.formula == "torch" else K.ADAM_KERAS) | \
                        (K.ADAM_DECOUPLED_WD if s.decoupled_weight_decay else 0)
        else:
            self.hyper = K.OptHyper(dev, lr=s.learning_rate, momentum=s.momentum, weight_decay=s.weight_decay,
                                    grad_scale=grad_scale)
            self.mode = (K.SGD_TORCH if s.formula == "torch" else K.SGD_KERAS) | (K.SGD_NESTEROV if s.nesterov else 0)
        self.ranges = [(0, flat.total)] if state_ranges is None else [tuple(r) for r in state_ranges]
        # moment storage is compact over the owned ranges
        self._base = []
        n = 0
        for a, b in self.ranges:
            self._base.append(n)
            n += b - a
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev) if s.kind == "adam" else None
        self.steps = 0

    # --------------------------------------------------------------- control
    def set_lr(self, lr: float) -> None:
        self.hyper.set(lr=lr)

    def set_grad_scale(self, s: float) -> None:
        self.hyper.set(grad_scale=s)

    def _locate(self, a: int, b: int) -> int:
        for (ra, rb), base in zip(self.ranges, self._base):
            if ra <= a and b <= rb:
                return base + (a - ra)
        raise ValueError(f"range [{a}, {b}) is not owned by this optimizer instance")

    @torch.no_grad()
    def step(self, ranges: Sequence[tuple[int, int]] | None = None, grad: torch.Tensor | None = None,
             params=None, tick: bool = True) -> None:
        """One optimizer step over ``ranges`` (default: everything owned). ``grad`` overrides the
        arena the gradients are read from (same indexing as ``flat.grad``). ``params`` overrides the fp32
        arena that is updated (same indexing as ``flat.master``; the bf16 shadow is then left alone): a
        parameter-server owner keeps its authoritative shard apart from the replica's working copy.
        ``tick=False``: a further call of the SAME optimizer step (one call per bucket): the step counter of the
        bias correction does not advance."""
        f = self.flat
        g_arena = f.grad if grad is None else grad
        if tick:
            self.hyper.tick()
            self.steps += 1
        for a, b in (self.ranges if ranges is None else ranges):
            if b <= a:
                continue
            o = self._locate(a, b)
            n = b - a
            p = f.m32(a, b) if params is None else params[a:b]
            g = g_arena[a:b]
            sh = f.shadow[a:b] if (f.shadow is not None and params is None) else None
            if self.spec.kind == "adam":
                K.adam_apply(p, g, self.m[o:o + n], self.v[o:o + n], self.hyper, shadow=sh, mode=self.mode)
            else:
                K.sgdm_apply(p, g, self.m[o:o + n], self.hyper, shadow=sh, mode=self.mode)

    # ------------------------------------------------------------ checkpoint
    def state_dict(self) -> dict:
        sd = {"kind": self.spec.kind, "steps": self.steps, "ranges": list(self.ranges),
              "m": self.m.detach().cpu().clone(), **self.hyper.state_dict()}
        if self.v is not None:
            sd["v"] = self.v.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd: dict) -> None:
        if [tuple(r) for r in sd["ranges"]] != [tuple(r) for r in self.ranges]:
            raise ValueError("optimizer state was saved for a different shard layout")
        self.m.copy_(sd["m"].to(self.m.device))
        if self.v is not None:
            self.v.copy_(sd["v"].to(self.v.device))
        self.hyper.load_state_dict(sd)
        self.steps = int(sd["steps"])
