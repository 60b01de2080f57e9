"""Keras-style layers on torch modules.

A user ``model_fn`` written against the reference looks like
``Sequential([Flatten(input_shape=(28, 28)), Dense(512, activation='relu'),
Dropout(0.2), Dense(10, activation='softmax')])`` (reference
initializer.py:12-21). These layers let that definition carry over unchanged:
shapes are inferred at ``Sequential`` construction from ``input_shape``,
activations are given by name, tensors are NHWC at the API boundary.

Device layout: image activations are kept as torch ``NCHW``-shaped tensors in
``channels_last`` memory format, i.e. physically NHWC, the layout the
MI355X convolution and normalisation kernels consume; ``Flatten`` flattens in
NHWC order so Dense kernels line up with Keras weights.

Every layer implements ``keras_weights()`` / ``set_keras_weights()`` with the
Keras array layouts (Dense kernel ``(in, out)``, Conv2D kernel ``HWIO``,
BatchNormalization ``gamma, beta, moving_mean, moving_variance``).
"""
from __future__ import annotations

import math
from typing import Callable, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import conv as _conv
from ..ops import dense as _dense
from ..ops import norm as _norm
from ..ops import pool as _pool


def _to_np(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def _assign(p: torch.Tensor, a: np.ndarray) -> None:
    a = torch.from_numpy(np.ascontiguousarray(a))
    if tuple(a.shape) != tuple(p.shape):
        raise ValueError(f"weight shape mismatch: expected {tuple(p.shape)}, got {tuple(a.shape)}")
    with torch.no_grad():
        p.copy_(a.to(p.dtype))


_ACTS: dict[str | None, Callable | None] = {
    None: None, "linear": None, "relu": F.relu, "gelu": F.gelu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
    "softmax": "softmax",
}


def _resolve_act(a):
    if callable(a):
        return a, getattr(a, "__name__", "custom")
    if a not in _ACTS:
        raise ValueError(f"unknown activation {a!r}")
    return _ACTS[a], a or "linear"


def glorot_uniform_(w: torch.Tensor, fan_in: int, fan_out: int) -> None:
    """Keras' default kernel initializer."""
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        w.uniform_(-lim, lim)


class Layer(nn.Module):
    """Base: ``build(in_shape) -> out_shape`` with shapes excluding the batch dim, NHWC order."""

    def build(self, in_shape: tuple) -> tuple:
        return in_shape

    def keras_weights(self) -> list[np.ndarray]:
        return []

    def set_keras_weights(self, ws: Sequence[np.ndarray]) -> int:
        """Consumes arrays from the front of ``ws``; returns how many were used."""
        return 0


class InputLayer(Layer):
    def __init__(self, input_shape):
        super().__init__()
        self.input_shape = tuple(input_shape)

    def forward(self, x):
        return x


class Flatten(Layer):
    def __init__(self, input_shape=None):
        super().__init__()
        self.input_shape = tuple(input_shape) if input_shape is not None else None

    def build(self, in_shape):
        return (int(np.prod(in_shape)),)

    def forward(self, x):
        if x.dim() == 4:
            x = x.permute(0, 2, 3, 1)  # NHWC element order (a free view for channels_last tensors)
        return x.reshape(x.shape[0], -1)


class Dense(Layer):
    """``y = act(x @ kernel + bias)``; bias add and ReLU/GELU run in the GEMM epilogue on GPU."""

    def __init__(self, units: int, activation=None, use_bias: bool = True, input_shape=None):
        super().__init__()
        self.units = units
        self.act, self.act_name = _resolve_act(activation)
        self.use_bias = use_bias
        self.input_shape = tuple(input_shape) if input_shape is not None else None
        self.weight = None
        self.bias = None

    def build(self, in_shape):
        fan_in = int(in_shape[-1])
        self.weight = nn.Parameter(torch.empty(self.units, fan_in))  # torch (out, in)
        glorot_uniform_(self.weight, fan_in, self.units)
        self.bias = nn.Parameter(torch.zeros(self.units)) if self.use_bias else None
        return tuple(in_shape[:-1]) + (self.units,)

    @property
    def is_softmax(self) -> bool:
        return self.act == "softmax"

    def forward(self, x, logits: bool = False, dropout: float = 0.0):
        """``dropout`` > 0 (set by ``Sequential`` when a ``Dropout`` layer follows) fuses the dropout
        into this layer's GEMM epilogue."""
        fused = self.act_name if self.act_name in ("relu", "gelu") else None
        if dropout and not (fused or self.act is None):
            raise ValueError("dropout can only be fused behind a linear, relu or gelu Dense layer")
        y = _dense.linear(x, self.weight, self.bias, activation=fused, dropout_p=dropout, training=self.training)
        if fused or self.act is None:
            return y
        if self.act == "softmax":
            return y if logits else torch.softmax(y.float(), dim=-1).to(y.dtype)
        return self.act(y)

    def keras_weights(self):
        ws = [_to_np(self.weight).T.copy()]
        if self.bias is not None:
            ws.append(_to_np(self.bias))
        return ws

    def set_keras_weights(self, ws):
        _assign(self.weight, np.asarray(ws[0]).T)
        if self.bias is not None:
            _assign(self.bias, np.asarray(ws[1]))
            return 2
        return 1


class Dropout(Layer):
    """On a GPU the mask comes from the framework's device-resident Philox state (``ops/rng.py``) and is
    regenerated, not stored, for the backward pass; ``Sequential`` fuses a ``Dropout`` that follows a ``Dense``
    layer into that layer's GEMM epilogue, so this module only runs on its own behind other layers."""

    def __init__(self, rate: float):
        super().__init__()
        self.rate = rate

    def forward(self, x):
        return _dense.dropout(x, self.rate, self.training)


class Activation(Layer):
    def __init__(self, activation):
        super().__init__()
        self.act, self.act_name = _resolve_act(activation)

    def forward(self, x):
        if self.act == "softmax":
            return torch.softmax(x.float(), dim=-1).to(x.dtype)
        return x if self.act is None else self.act(x)


class ReLU(Activation):
    def __init__(self):
        super().__init__("relu")


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _to_device_layout(x: torch.Tensor) -> torch.Tensor:
    """NHWC-shaped input -> NCHW-shaped channels_last tensor (a view)."""
    return x.permute(0, 3, 1, 2)


class Conv2D(Layer):
    """2-D convolution, Keras semantics (NHWC API, ``padding`` 'valid' | 'same')."""

    def __init__(self, filters: int, kernel_size, strides=1, padding: str = "valid", activation=None,
                 use_bias: bool = True, input_shape=None):
        super().__init__()
        self.filters = filters
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = padding.lower()
        if self.padding not in ("valid", "same"):
            raise ValueError("padding must be 'valid' or 'same'")
        self.act, self.act_name = _resolve_act(activation)
        self.use_bias = use_bias
        self.input_shape = tuple(input_shape) if input_shape is not None else None
        self.weight = None
        self.bias = None

    def _pads(self, h, w):
        if self.padding == "valid":
            return (0, 0, 0, 0)
        kh, kw = self.kernel_size
        sh, sw = self.strides
        ph = max((math.ceil(h / sh) - 1) * sh + kh - h, 0)
        pw = max((math.ceil(w / sw) - 1) * sw + kw - w, 0)
        return (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)  # left, right, top, bottom (TF puts the extra at the end)

    def build(self, in_shape):
        h, w, c = in_shape
        kh, kw = self.kernel_size
        self.weight = nn.Parameter(torch.empty(self.filters, c, kh, kw).contiguous(memory_format=torch.channels_last))
        glorot_uniform_(self.weight, c * kh * kw, self.filters * kh * kw)
        self.bias = nn.Parameter(torch.zeros(self.filters)) if self.use_bias else None
        l, r, t, b = self._pads(h, w)
        oh = (h + t + b - kh) // self.strides[0] + 1
        ow = (w + l + r - kw) // self.strides[1] + 1
        return (oh, ow, self.filters)

    def forward(self, x):
        pads = self._pads(x.shape[2], x.shape[3])
        fused = "relu" if self.act_name == "relu" else None      # bias and ReLU ride the GEMM epilogue
        y = _conv.conv2d_bias_act(x, self.weight, self.bias, self.strides, pads, fused)
        if fused or self.act is None:
            return y
        if self.act == "softmax":
            return torch.softmax(y.float(), dim=1).to(y.dtype)
        return self.act(y)

    def keras_weights(self):
        ws = [np.ascontiguousarray(_to_np(self.weight).transpose(2, 3, 1, 0))]  # OIHW -> HWIO
        if self.bias is not None:
            ws.append(_to_np(self.bias))
        return ws

    def set_keras_weights(self, ws):
        _assign(self.weight, np.asarray(ws[0]).transpose(3, 2, 0, 1))
        if self.bias is not None:
            _assign(self.bias, np.asarray(ws[1]))
            return 2
        return 1


class _Pool2D(Layer):
    def __init__(self, pool_size=2, strides=None, padding="valid"):
        super().__init__()
        self.pool_size = _pair(pool_size)
        self.strides = _pair(strides) if strides is not None else self.pool_size
        if padding.lower() != "valid":
            raise ValueError("pooling supports padding='valid'")

    def build(self, in_shape):
        h, w, c = in_shape
        return ((h - self.pool_size[0]) // self.strides[0] + 1, (w - self.pool_size[1]) // self.strides[1] + 1, c)


class MaxPooling2D(_Pool2D):
    def forward(self, x):
        k, s = self.pool_size, self.strides
        if k[0] == k[1] and s[0] == s[1]:
            return _pool.max_pool2d(x, k[0], s[0])
        if x.is_cuda:
            from ..ops import generic_nn as _gen
            return _gen.max_pool2d(x, k, s)
        return F.max_pool2d(x, k, s)


class AveragePooling2D(_Pool2D):
    def forward(self, x):
        return _pool.avg_pool2d(x, self.pool_size, self.strides)


class GlobalAveragePooling2D(Layer):
    def build(self, in_shape):
        return (in_shape[-1],)

    def forward(self, x):
        from ..ops import pool as _pool
        return _pool.global_avg_pool(x)


class BatchNormalization(Layer):
    """Batch normalisation over the channel axis (Keras defaults momentum 0.99, epsilon 1e-3)."""

    def __init__(self, momentum: float = 0.99, epsilon: float = 1e-3):
        super().__init__()
        self.momentum, self.epsilon = momentum, epsilon
        self.bn = None

    def build(self, in_shape):
        c = in_shape[-1]
        cls = nn.BatchNorm2d if len(in_shape) == 3 else nn.BatchNorm1d
        self.bn = cls(c, eps=self.epsilon, momentum=1.0 - self.momentum)
        return in_shape

    def forward(self, x):
        bn = self.bn
        if x.dim() == 4 and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 \
                and bn.weight.dtype == torch.float32:
            if not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)
            return _norm.batch_norm_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training,
                                        bn.momentum, bn.eps, relu=False)
        if x.is_cuda and x.dim() in (2, 4) and x.dtype in (torch.float32, torch.bfloat16):
            # every other GPU case (fp32, any width, [N, C] inputs): the framework's general kernels
            from ..ops import generic_nn as _gen
            if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
            return _gen.batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, bn.momentum,
                                   bn.eps)
        return bn(x)

    def keras_weights(self):
        return [_to_np(self.bn.weight), _to_np(self.bn.bias), _to_np(self.bn.running_mean), _to_np(self.bn.running_var)]

    def set_keras_weights(self, ws):
        for p, a in zip((self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var), ws[:4]):
            _assign(p, np.asarray(a))
        return 4


class Embedding(Layer):
    def __init__(self, input_dim: int, output_dim: int):
        super().__init__()
        self.emb = nn.Embedding(input_dim, output_dim)
        with torch.no_grad():
            self.emb.weight.uniform_(-0.05, 0.05)

    def build(self, in_shape):
        return tuple(in_shape) + (self.emb.embedding_dim,)

    def forward(self, x):
        if self.emb.weight.is_cuda:
            from ..ops import embedding as _emb
            return _emb.embedding(self.emb.weight, x)      # own gather / sorted-run gradient kernels
        return self.emb(x)

    def keras_weights(self):
        return [_to_np(self.emb.weight)]

    def set_keras_weights(self, ws):
        _assign(self.emb.weight, np.asarray(ws[0]))
        return 1
