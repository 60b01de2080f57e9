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
    """1x1 convolution, stride s >= 1, as three GEMMs on the NHWC bytes. With a stride the GEMM
    reads (forward, weight gradient) or writes (data gradient) only the strided pixels through the
    kernel's row map, so no gathered copy of the activation is ever made."""

    @staticmethod
    def forward(ctx, x, w, stride):
        N, C, H, W = x.shape
        K = w.shape[0]
        OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
        w2 = w.reshape(K, C)
        y = torch.empty((N, OH, OW, K), dtype=x.dtype, device=x.device)
        ex = _gemm.pixel_map(_gemm.MAP_A, (OH, OW), (H, W), stride) if stride != 1 else None
        _gemm.gemm_raw(_gemm.NT, x, w2, y, N * OH * OW, K, C, C, C, K, extra=ex)
        ctx.save_for_backward(x, w2)
        ctx.dims = (N, C, H, W, K, OH, OW, stride)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x, w2 = ctx.saved_tensors
        N, C, H, W, K, OH, OW, stride = ctx.dims
        if not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.contiguous(memory_format=torch.channels_last)
        M = N * OH * OW
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if stride != 1:
                gx = torch.zeros((N, H, W, C), dtype=x.dtype, device=x.device)
                ex = _gemm.pixel_map(_gemm.MAP_C, (OH, OW), (H, W), stride)
            else:
                gx = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
                ex = None
            _gemm.gemm_raw(_gemm.NN, gy, w2, gx, M, C, K, K, C, C, extra=ex)
            gx = gx.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = torch.empty((K, C), dtype=w2.dtype, device=x.device)
            ex = _gemm.pixel_map(_gemm.MAP_B, (OH, OW), (H, W), stride) if stride != 1 else None
            _gemm.gemm_raw(_gemm.TN, gy, x, gw, K, C, M, K, C, C, extra=ex)
            gw = gw.view(K, 1, 1, C).permute(0, 3, 1, 2)   # [K, C, 1, 1], channels_last
        return gx, gw, None


def _gemm_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.shape[1] % 8 == 0
            and w.shape[0] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last))


def conv2d(x: torch.Tensor, w: torch.Tensor, stride: int = 1, padding: int = 0) -> torch.Tensor:
    """Bias-free convolution of a channels_last activation with an ``[O, I, kh, kw]`` weight."""
    if w.shape[2] == 1 and w.shape[3] == 1 and padding == 0 and _backend == "gemm" and _gemm_ok(x, w):
        return _Conv1x1Fn.apply(x, w, stride)
    if x.dtype != w.dtype:
        x = x.to(w.dtype)
    return F.conv2d(x, w, None, stride, padding)
