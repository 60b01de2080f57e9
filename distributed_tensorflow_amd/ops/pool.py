"""Max pooling and global average pooling of channels_last (NHWC) activations.

bf16 GPU tensors with a channel count that is a multiple of 8 run on the 16-byte kernels of
``csrc/kernels/pool.hip`` (forward records the winning tap in one byte per element; backward is
an atomics-free gather); any other GPU tensor (fp32, odd channel counts, rectangular windows) on the general kernels
of ``csrc/kernels/generic_nn.hip``; only CPU tensors use ``torch.nn.functional``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib
from . import generic_nn as _gen


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        N, C, H, W = x.shape
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = torch.empty((N, OH, OW, C), dtype=x.dtype, device=x.device)
        idx = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
        rc = _lib.kernels().dtf_maxpool_fwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(idx), N, H, W, C, OH, OW, k, s, p,
                                            _lib.stream_ptr(x.device))
        _lib.check(rc, "maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.geom = (N, C, H, W, OH, OW, k, s, p)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        N, C, H, W, OH, OW, k, s, p = ctx.geom
        if not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.contiguous(memory_format=torch.channels_last)
        gx = torch.empty((N, H, W, C), dtype=gy.dtype, device=gy.device)
        rc = _lib.kernels().dtf_maxpool_bwd(_lib.ptr(gy), _lib.ptr(idx), _lib.ptr(gx), N, H, W, C, OH, OW, k, s, p,
                                            _lib.stream_ptr(gy.device))
        _lib.check(rc, "maxpool_bwd")
        return gx.permute(0, 3, 1, 2), None, None, None


def max_pool2d(x: torch.Tensor, kernel_size: int, stride: int | None = None, padding: int = 0) -> torch.Tensor:
    s = kernel_size if stride is None else stride
    if (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0 and padding < kernel_size
            and x.is_contiguous(memory_format=torch.channels_last)):
        return _MaxPoolFn.apply(x, kernel_size, s, padding)
    if _gen.usable(x) and x.dim() == 4:
        return _gen.max_pool2d(x, kernel_size, s, padding)
    return F.max_pool2d(x, kernel_size, s, padding)


def avg_pool2d(x: torch.Tensor, kernel_size, stride=None, padding=0) -> torch.Tensor:
    if _gen.usable(x) and x.dim() == 4:
        return _gen.avg_pool2d(x, kernel_size, stride, padding)
    return F.avg_pool2d(x, kernel_size, stride, padding)


class _GlobalAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        y = torch.empty((N, C), dtype=x.dtype, device=x.device)
        rc = _lib.kernels().dtf_gap_fwd(_lib.ptr(x), _lib.ptr(y), N, H * W, C, _lib.stream_ptr(x.device))
        _lib.check(rc, "gap_fwd")
        ctx.geom = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.geom
        gy = gy.contiguous()
        gx = torch.empty((N, H, W, C), dtype=gy.dtype, device=gy.device)
        rc = _lib.kernels().dtf_gap_bwd(_lib.ptr(gy), _lib.ptr(gx), N, H * W, C, _lib.stream_ptr(gy.device))
        _lib.check(rc, "gap_bwd")
        return gx.permute(0, 3, 1, 2)


def global_avg_pool(x: torch.Tensor) -> torch.Tensor:
    """``x.mean(dim=(2, 3))`` of an ``[N, C, H, W]`` activation: the head of the ResNets. bf16 channels_last GPU
    tensors run on ``gap_fwd_kernel`` / ``gap_bwd_kernel`` (fp32 accumulation; the backward pass is one streaming
    write instead of an expand, a copy and a division)."""
    if (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last)):
        return _GlobalAvgPoolFn.apply(x)
    if _gen.usable(x) and x.dim() == 4 and x.shape[0] <= 65535:
        return _gen.global_avg_pool(x)
    return x.mean(dim=(2, 3))
