"""2-D convolution front end for channels_last (NHWC) activations.

Routing:
* 1x1 convolutions (two thirds of ResNet-50's layers) ARE matrix products on NHWC data:
  ``y[N*H*W, Cout] = x[N*H*W, Cin] @ w[Cout, Cin]^T``. With bf16 operands they run on the
  hand-written MFMA GEMM (``ops/gemm.py``): forward, data gradient and the split-K weight
  gradient, with no im2col and no layout change. A stride-2 1x1 convolution first gathers
  the strided pixels.
* everything else goes to the vendor convolution library through ``torch.nn.functional``.

``set_conv1x1_backend('gemm' | 'library')`` switches the 1x1 route (default from the
``DTF_CONV1X1`` environment variable, ``gemm``).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import gemm as _gemm

_backend = os.environ.get("DTF_CONV1X1", "gemm")


def set_conv1x1_backend(name: str) -> None:
    global _backend
    if name not in ("gemm", "library"):
        raise ValueError("backend must be 'gemm' or 'library'")
    _backend = name


def get_conv1x1_backend() -> str:
    return _backend


class _Conv1x1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        N, C, H, W = x.shape
        K = w.shape[0]
        x2 = x.permute(0, 2, 3, 1).reshape(N * H * W, C)          # view of the NHWC bytes
        w2 = w.reshape(K, C)
        y2 = _gemm.gemm_nt(x2, w2)
        ctx.save_for_backward(x2, w2)
        ctx.dims = (N, C, H, W, K)
        ctx.w_shape, ctx.w_stride = w.shape, w.stride()
        return y2.view(N, H, W, K).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x2, w2 = ctx.saved_tensors
        N, C, H, W, K = ctx.dims
        g2 = gy.permute(0, 2, 3, 1).reshape(N * H * W, K)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _gemm.gemm_nn(g2, w2).view(N, H, W, C).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = _gemm.gemm_tn(g2, x2).view(K, 1, 1, C).permute(0, 3, 1, 2)   # [K, C, 1, 1], channels_last
        return gx, gw


def _gemm_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.shape[1] % 8 == 0
            and w.shape[0] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last))


def conv2d(x: torch.Tensor, w: torch.Tensor, stride: int = 1, padding: int = 0) -> torch.Tensor:
    """Bias-free convolution of a channels_last activation with an ``[O, I, kh, kw]`` weight."""
    if w.shape[2] == 1 and w.shape[3] == 1 and padding == 0 and _backend == "gemm" and _gemm_ok(x, w):
        if stride != 1:
            x = x[:, :, ::stride, ::stride].contiguous(memory_format=torch.channels_last)
        return _Conv1x1Fn.apply(x, w)
    if x.dtype != w.dtype:
        x = x.to(w.dtype)
    return F.conv2d(x, w, None, stride, padding)
