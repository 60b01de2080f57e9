"""ResNet-20 (CIFAR-10) and ResNet-50 (ImageNet), NHWC / bf16 first.

The reference ships only an MLP (reference initializer.py:12-21); these are the
convolutional models of the named benchmark configurations (CIFAR-10 ResNet-20
under the asynchronous parameter server, ResNet-50 under the mirrored
all-reduce strategy).

Layout: activations are ``channels_last`` (physically NHWC) end to end and
convolution weights are stored ``channels_last`` too, so no layout transposes
are inserted around the convolutions. Every convolution is followed by
``BNAct``: batch normalisation fused with the optional residual add and ReLU
into two memory passes forward and two backward (``ops/norm.py``), instead of
separate normalise / add / activation kernels.

Keras weight export order: per layer ``kernel (HWIO)`` then ``gamma, beta,
moving_mean, moving_variance``, as ``keras.applications``-style models list them.
"""
from __future__ import annotations


import numpy as np
import torch
import torch.nn as nn

from ..api.model import Model
from ..ops import conv as conv_ops
from ..ops import dense as dense_ops
from ..ops import norm as norm_ops
from ..ops import pool as pool_ops


class Conv(nn.Module):
    """Bias-free 2-D convolution with channels_last weights (He-normal init, fan-out)."""

    def __init__(self, cin: int, cout: int, k: int, stride: int = 1, padding: int | None = None):
        super().__init__()
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.padding = (k // 2) if padding is None else padding
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))

    def forward(self, x, bn_stats: bool = False):
        """``bn_stats=True``: ``(y, partial batch-norm statistics of y or None)``, see ``ops.conv.conv2d``."""
        return conv_ops.conv2d(x, self.weight, self.stride, self.padding, bn_stats)

    def keras_weights(self):
        return [np.ascontiguousarray(self.weight.detach().float().cpu().numpy().transpose(2, 3, 1, 0))]

    def set_keras_weights(self, ws):
        a = torch.from_numpy(np.ascontiguousarray(np.asarray(ws[0]).transpose(3, 2, 0, 1)))
        with torch.no_grad():
            self.weight.copy_(a.to(self.weight.dtype))
        return 1


class BNAct(nn.Module):
    """Batch normalisation (+ residual add) (+ ReLU) over the channel axis; parameters and running
    statistics stay fp32 whatever the activation dtype."""
    keep_fp32 = True

    def __init__(self, c: int, relu: bool = True, eps: float = 1e-5, momentum: float = 0.1, zero_init: bool = False):
        super().__init__()
        self.c, self.relu, self.eps, self.momentum = c, relu, eps, momentum
        self.weight = nn.Parameter(torch.zeros(c) if zero_init else torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))

    def forward(self, x, residual=None, partials=None):
        return norm_ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                                       self.training, self.momentum, self.eps, self.relu, residual,
                                       partials if self.training else None)

    def deferred(self, x, partials):
        """The norm with its apply pass left to the convolution that consumes it (``ops.norm.DeferredNorm``) when
        that is possible (training, statistics handed over by the producer, bf16 GPU fast path), else the tensor."""
        if self.training and partials is not None:
            d = norm_ops.batch_norm_deferred(x, self.weight, self.bias, self.running_mean, self.running_var,
                                             self.momentum, self.eps, self.relu, partials)
            if d is not None:
                return d
        return self.forward(x, partials=partials)

    def lazy(self, x, partials):
        """A plain norm (no ReLU) that only feeds the residual input of a norm-add-ReLU: statistics now, the apply
        inside that norm's kernel (``ops.norm.batch_norm_lazy``) when possible, else the tensor."""
        if self.training and partials is not None and not self.relu:
            d = norm_ops.batch_norm_lazy(x, self.weight, self.bias, self.running_mean, self.running_var,
                                         self.momentum, self.eps, partials)
            if d is not None:
                return d
        return self.forward(x, partials=partials)

    def keras_weights(self):
        return [t.detach().float().cpu().numpy() for t in (self.weight, self.bias, self.running_mean, self.running_var)]

    def set_keras_weights(self, ws):
        with torch.no_grad():
            for t, a in zip((self.weight, self.bias, self.running_mean, self.running_var), ws[:4]):
                t.copy_(torch.from_numpy(np.asarray(a)).to(t.dtype))
        return 4


def _conv_bn(conv: "Conv", bn: "BNAct", x, residual=None, lazy: bool = False):
    """``bn(conv(x) (+ residual))``; in training the convolution hands the batch norm the partial statistics it
    gathered while storing its output, so the norm reads that output once instead of twice. ``lazy``: the result only
    feeds the residual input of the block's closing norm-add-ReLU (see :meth:`BNAct.lazy`)."""
    if bn.training:
        y, st = conv(x, bn_stats=True)
        if lazy:
            return bn.lazy(y, st)
        return bn(y, residual=residual, partials=st)
    return bn(conv(x), residual=residual)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin: int, width: int, stride: int):
        super().__init__()
        cout = width * 4
        self.conv1, self.bn1 = Conv(cin, width, 1), BNAct(width)
        self.conv2, self.bn2 = Conv(width, width, 3, stride), BNAct(width)
        self.conv3, self.bn3 = Conv(width, cout, 1), BNAct(cout, relu=True, zero_init=True)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = Conv(cin, cout, 1, stride)
            self.down_bn = BNAct(cout, relu=False)

    def forward(self, x):
        # the alias xi carries the identity branch's gradient into conv1's data-gradient epilogue
        c1 = self.conv1
        if self.training:
            y, xi, st = conv_ops.conv2d_fork(x, c1.weight, c1.stride, c1.padding, bn_stats=True)
        else:
            (y, xi), st = conv_ops.conv2d_fork(x, c1.weight, c1.stride, c1.padding), None
        idt = xi if self.down is None else _conv_bn(self.down, self.down_bn, xi, lazy=True)
        # bn1 and bn2 have one consumer each, the next convolution: it normalises while it stages its operand, so
        # their normalised outputs are never written
        y = self.bn1.deferred(y, st)
        if self.training:
            y, st = self.conv2(y, bn_stats=True)
            y = self.bn2.deferred(y, st)
        else:
            y = self.bn2(self.conv2(y))
        return _conv_bn(self.conv3, self.bn3, y, residual=idt)   # relu(bn(conv) + identity) in one fused pass


class BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1, self.bn1 = Conv(cin, cout, 3, stride), BNAct(cout)
        self.conv2, self.bn2 = Conv(cout, cout, 3), BNAct(cout, relu=True)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = Conv(cin, cout, 1, stride)
            self.down_bn = BNAct(cout, relu=False)

    def forward(self, x):
        # as in Bottleneck: the alias xi carries the identity (or down-sampling) branch's gradient into conv1's
        # data-gradient epilogue, so autograd runs no add kernel over the block input and the norm before this
        # block sees a single consumer (its backward sums then come from that epilogue)
        c1 = self.conv1
        if self.training:
            y, xi, st = conv_ops.conv2d_fork(x, c1.weight, c1.stride, c1.padding, bn_stats=True)
            y = self.bn1(y, partials=st)
        else:
            y, xi = conv_ops.conv2d_fork(x, c1.weight, c1.stride, c1.padding)
            y = self.bn1(y)
        idt = xi if self.down is None else _conv_bn(self.down, self.down_bn, xi, lazy=True)
        return _conv_bn(self.conv2, self.bn2, y, residual=idt)


class Classifier(nn.Module):
    """Global average pool + dense layer producing logits (softmax is fused into the loss)."""

    def __init__(self, cin: int, classes: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(classes, cin))
        nn.init.normal_(self.weight, std=0.01)
        self.bias = nn.Parameter(torch.zeros(classes))

    def forward(self, x):
        x = pool_ops.global_avg_pool(x)
        return dense_ops.linear(x, self.weight, self.bias)

    def keras_weights(self):
        return [self.weight.detach().float().cpu().numpy().T.copy(), self.bias.detach().float().cpu().numpy()]

    def set_keras_weights(self, ws):
        with torch.no_grad():
            self.weight.copy_(torch.from_numpy(np.asarray(ws[0]).T.copy()).to(self.weight.dtype))
            self.bias.copy_(torch.from_numpy(np.asarray(ws[1])).to(self.bias.dtype))
        return 2


class ResNet50(Model):
    supports_logits = True
    input_layout = "NHWC"

    def __init__(self, classes: int = 1000, layers=(3, 4, 6, 3), width: int = 64):
        super().__init__()
        self.stem, self.stem_bn = Conv(3, width, 7, 2, 3), BNAct(width)
        blocks = []
        cin = width
        for i, n in enumerate(layers):
            w = width * (2 ** i)
            for j in range(n):
                blocks.append(Bottleneck(cin, w, 2 if (j == 0 and i > 0) else 1))
                cin = w * 4
        self.blocks = nn.Sequential(*blocks)
        self.head = Classifier(cin, classes)

    def forward(self, x, logits: bool = False):
        x = self.prepare_input(x)
        bn = self.stem_bn            # norm + ReLU + max pool in one pass over the widest activation
        z, st = self.stem(x, bn_stats=True)     # the stem convolution reduces the norm's statistics while it stores z
        x = norm_ops.batch_norm_relu_max_pool(z, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                              bn.training, bn.momentum, bn.eps, 3, 2, 1, st if bn.training else None)
        x = self.blocks(x)
        y = self.head(x)
        return y if logits else torch.softmax(y.float(), dim=-1)


class ResNet20(Model):
    """The CIFAR-10 ResNet of He et al. (3 stages x 3 basic blocks, 16/32/64 channels, ~0.27 M parameters)."""
    supports_logits = True
    input_layout = "NHWC"

    def __init__(self, classes: int = 10, n: int = 3):
        super().__init__()
        self.stem, self.stem_bn = Conv(3, 16, 3), BNAct(16)
        blocks = []
        cin = 16
        for i, c in enumerate((16, 32, 64)):
            for j in range(n):
                blocks.append(BasicBlock(cin, c, 2 if (j == 0 and i > 0) else 1))
                cin = c
        self.blocks = nn.Sequential(*blocks)
        self.head = Classifier(cin, classes)

    def forward(self, x, logits: bool = False):
        x = self.prepare_input(x)
        x = _conv_bn(self.stem, self.stem_bn, x)
        x = self.blocks(x)
        y = self.head(x)
        return y if logits else torch.softmax(y.float(), dim=-1)


def resnet50(classes: int = 1000) -> ResNet50:
    return ResNet50(classes)


def resnet20(classes: int = 10) -> ResNet20:
    return ResNet20(classes)
