"""The ``model_fn`` side of the user contract.

The reference calls exactly these members on whatever ``model_fn()`` returns:
the forward call (reference centralized/client.py:12), ``trainable_variables``
(client.py:19, server.py:92), ``get_weights()`` / ``set_weights()``
(server.py:96,150; client.py:72,90) and, for the mirrored strategy,
``compile`` / ``fit`` / ``evaluate`` (decentralized/native/dist_keras.py:30-32,
42,53). :class:`Model` provides them on top of ``torch.nn.Module``;
:class:`Sequential` adds Keras-style shape inference; :func:`as_model` adapts
a plain ``torch.nn.Module``.

Weight lists are in Keras order and layout (see ``api/keras_weights.py``), so a
list exported by Keras for the same architecture can be passed to
``set_weights`` and vice versa.
"""
from __future__ import annotations

from typing import Iterable, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import keras_weights as kw
from .layers import Dense, Dropout, Layer, _to_device_layout


class Model(nn.Module):
    """Base class: a torch module with the Keras-flavoured members the strategies call."""

    #: True when ``forward(x, logits=True)`` returns pre-softmax logits
    supports_logits = False
    #: expected rank-4 input layout at the API boundary
    input_layout = "NHWC"

    def __init__(self):
        super().__init__()
        self._compiled = None

    # ------------------------------------------------------------- variables
    @property
    def trainable_variables(self) -> list[nn.Parameter]:
        return [p for p in self.parameters() if p.requires_grad]

    def get_weights(self) -> list[np.ndarray]:
        return kw.get_weights(self)

    def set_weights(self, weights: Sequence[np.ndarray]) -> None:
        kw.set_weights(self, weights)

    def save_weights(self, path: str) -> None:
        kw.save_weights(self, path)

    def load_weights(self, path: str) -> None:
        kw.load_weights(self, path)

    def count_params(self) -> int:
        return sum(p.numel() for p in self.parameters())

    # ------------------------------------------------ keras training facade
    def compile(self, optimizer="adam", loss="sparse_categorical_crossentropy", metrics=("accuracy",)) -> None:
        """Records the training configuration; the strategy that later runs ``fit`` builds the
        fused optimizer over its flat arenas from it."""
        self._compiled = {"optimizer": optimizer, "loss": loss, "metrics": tuple(metrics or ())}

    def fit(self, x, epochs: int = 1, verbose: int = 0, strategy=None, **kw_):
        from ..parallel.mirrored import fit as _fit
        return _fit(self, x, epochs=epochs, verbose=verbose, strategy=strategy, **kw_)

    def evaluate(self, x, strategy=None, **kw_):
        from ..parallel.mirrored import evaluate as _evaluate
        return _evaluate(self, x, strategy=strategy, **kw_)

    # ---------------------------------------------------------------- inputs
    def prepare_input(self, x: torch.Tensor) -> torch.Tensor:
        """API-boundary tensor (NHWC images, any float dtype) -> the layout/dtype the layers compute in."""
        p = next(self.parameters(), None)
        if p is not None and x.is_floating_point() and x.dtype != p.dtype:
            x = x.to(p.dtype)
        shape = getattr(self, "input_shape_", None)
        if shape is not None and len(shape) == 3 and shape[-1] == 1 and x.dim() == 3:
            x = x.unsqueeze(-1)      # grey-scale images fed without their channel axis ([N, H, W])
        if x.dim() == 4 and self.input_layout == "NHWC":
            x = _to_device_layout(x)
        return x


class Sequential(Model):
    def __init__(self, layers: Iterable[nn.Module], input_shape=None):
        super().__init__()
        layers = list(layers)
        self.layers = nn.ModuleList(layers)
        shape = input_shape
        if shape is None and layers:
            shape = getattr(layers[0], "input_shape", None)
        self.input_shape_ = tuple(shape) if shape is not None else None
        if self.input_shape_ is not None:
            self.build(self.input_shape_)
        last = layers[-1] if layers else None
        self.supports_logits = isinstance(last, Dense) and last.is_softmax

    def build(self, input_shape) -> tuple:
        s = tuple(input_shape)
        for l in self.layers:
            if isinstance(l, Layer):
                s = l.build(s)
        self.output_shape_ = s
        return s

    def forward(self, x, logits: bool = False):
        x = self.prepare_input(x)
        n = len(self.layers)
        skip = False
        for i, l in enumerate(self.layers):
            if skip:            # a Dropout already applied inside the preceding Dense
                skip = False
                continue
            if i == n - 1 and logits and self.supports_logits:
                x = l(x, logits=True)
            elif (isinstance(l, Dense) and i + 1 < n and isinstance(self.layers[i + 1], Dropout)
                  and (l.act is None or l.act_name in ("relu", "gelu"))):
                x = l(x, dropout=self.layers[i + 1].rate)
                skip = True
            else:
                x = l(x)
        return x


class _Adapted(Model):
    def __init__(self, module: nn.Module, supports_logits: bool, input_layout: str):
        super().__init__()
        self.module = module
        self.supports_logits = supports_logits
        self.input_layout = input_layout

    def forward(self, x, logits: bool = False):
        x = self.prepare_input(x)
        if self.supports_logits:
            return self.module(x, logits=logits)
        return self.module(x)


def as_model(obj, supports_logits: bool | None = None, input_layout: str = "NCHW") -> Model:
    """Wraps what a user ``model_fn`` returned so that it satisfies the contract."""
    if isinstance(obj, Model):
        return obj
    if isinstance(obj, nn.Module):
        if supports_logits is None:
            supports_logits = bool(getattr(obj, "supports_logits", False))
        return _Adapted(obj, supports_logits, input_layout)
    raise TypeError(f"model_fn must return a Model or torch.nn.Module, got {type(obj).__name__}")
