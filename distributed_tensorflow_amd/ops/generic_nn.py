"""Front end of ``csrc/kernels/generic_nn.hip``: batch norm and pooling for any element type (fp32, bf16) and any
channel count on a GPU.

``ops/norm.py`` and ``ops/pool.py`` route here what their bf16 / multiple-of-8-channels kernels do not take -- above
all fp32 models, the configuration of the reference (reference initializer.py:40, centralized/client.py:80-81) and the
launcher's default -- so that a GPU run of such a model contains no vendor-library normalisation or pooling kernel.
Activations are ``[N, C, H, W]`` tensors in channels_last memory (NHWC), or ``[N, C]`` matrices.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

_ws: dict = {}


def usable(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() in (2, 4)


def _scratch(dev, numel: int) -> torch.Tensor:
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    t = _ws.get(key)
    if t is None or t.numel() < numel:
        t = _ws[key] = torch.empty(max(numel, 1 << 16), dtype=torch.float32, device=dev)
    return t


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    """The tensor in the memory order the kernels read ([N, H, W, C] rows of channels), without a copy when it already is."""
    if x.dim() == 2:
        return x.contiguous()
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


def _f32(t):
    return None if t is None else (t if t.dtype == torch.float32 else t.float()).contiguous()


class _BatchNormFn(torch.autograd.Function):
    """Training-mode batch norm (+ residual add, + ReLU) with batch statistics; running statistics updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, momentum, eps, relu, residual):
        x = _nhwc(x)
        C = x.shape[1]
        M = x.numel() // C
        dev = x.device
        lib = _lib.kernels()
        g32, b32 = _f32(gamma), _f32(beta)
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)       # mean, invstd, scale, shift
        ws = _scratch(dev, lib.dtf_gbn_parts(ctypes.c_longlong(M), C) * 2 * C)
        rc = lib.dtf_gbn_fwd_stats(_lib.ptr(x), _lib.dtype_code(x.dtype), ctypes.c_longlong(M), C, _lib.ptr(g32),
                                   _lib.ptr(b32), _lib.ptr(run_mean), _lib.ptr(run_var), _lib.ptr(stats[0]),
                                   _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), ctypes.c_float(eps),
                                   ctypes.c_float(momentum), _lib.ptr(ws), _lib.stream_ptr(dev))
        _lib.check(rc, "gbn_fwd_stats")
        res = None
        if residual is not None:
            res = _nhwc(residual if residual.dtype == x.dtype else residual.to(x.dtype))
        y = torch.empty_like(x)
        rc = lib.dtf_gbn_apply_fwd(_lib.ptr(x), _lib.ptr(res), _lib.ptr(y), _lib.dtype_code(x.dtype), _lib.ptr(stats[2]),
                                   _lib.ptr(stats[3]), ctypes.c_longlong(x.numel()), C, int(relu), _lib.stream_ptr(dev))
        _lib.check(rc, "gbn_apply_fwd")
        ctx.save_for_backward(x, y if relu else None, g32, stats)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.param_dtypes = (gamma.dtype if gamma is not None else None, beta.dtype if beta is not None else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, g32, stats = ctx.saved_tensors
        C = x.shape[1]
        M = x.numel() // C
        dev = x.device
        lib = _lib.kernels()
        gy = _nhwc(gy if gy.dtype == x.dtype else gy.to(x.dtype))
        dx = torch.empty_like(x)
        gres = torch.empty_like(x) if (ctx.has_res and ctx.relu) else None
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        coef = torch.empty(3, C, dtype=torch.float32, device=dev)
        ws = _scratch(dev, lib.dtf_gbn_parts(ctypes.c_longlong(M), C) * 2 * C)
        rc = lib.dtf_gbn_bwd(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(y), _lib.ptr(dx), _lib.ptr(gres),
                             _lib.dtype_code(x.dtype), ctypes.c_longlong(M), C, _lib.ptr(g32), _lib.ptr(stats[0]),
                             _lib.ptr(stats[1]), _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(coef), int(ctx.relu),
                             _lib.ptr(ws), _lib.stream_ptr(dev))
        _lib.check(rc, "gbn_bwd")
        gdt, bdt = ctx.param_dtypes
        dres = None
        if ctx.has_res:
            dres = gres if gres is not None else gy
        return (dx, dgb[0].to(gdt) if gdt is not None else None, dgb[1].to(bdt) if bdt is not None else None, None, None,
                None, None, None, dres)


def batch_norm(x, gamma, beta, running_mean, running_var, training: bool, momentum: float = 0.1, eps: float = 1e-5,
               relu: bool = False, residual=None) -> torch.Tensor:
    """``act(batch_norm(x) + residual)`` over the channel axis (dim 1) of a GPU tensor of any float type and width."""
    if training:
        return _BatchNormFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual)
    C = x.shape[1]
    scale = torch.rsqrt(running_var.float() + eps)
    if gamma is not None:
        scale = scale * gamma.float()
    shift = -running_mean.float() * scale
    if beta is not None:
        shift = shift + beta.float()
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or (gamma is not None and gamma.requires_grad))
    if needs_grad:                      # inference-mode statistics inside a differentiated graph (rare): plain ops
        shp = (1, C) + (1,) * (x.dim() - 2)
        y = x * scale.to(x.dtype).view(shp) + shift.to(x.dtype).view(shp)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    xc = _nhwc(x)
    res = _nhwc(residual.to(x.dtype)) if residual is not None else None
    y = torch.empty_like(xc)
    rc = _lib.kernels().dtf_gbn_apply_fwd(_lib.ptr(xc), _lib.ptr(res), _lib.ptr(y), _lib.dtype_code(xc.dtype),
                                          _lib.ptr(scale.contiguous()), _lib.ptr(shift.contiguous()),
                                          ctypes.c_longlong(xc.numel()), C, int(relu), _lib.stream_ptr(xc.device))
    _lib.check(rc, "gbn_apply_fwd")
    return y


# ------------------------------------------------------------------------------------------------------- pooling
def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode, k, s, p):
        x = _nhwc(x)
        N, C, H, W = x.shape
        OH, OW = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
        y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty(N * OH * OW * C, dtype=torch.uint8, device=x.device) if mode == 0 else None
        rc = _lib.kernels().dtf_gpool(0, mode, _lib.dtype_code(x.dtype), _lib.ptr(x), _lib.ptr(y), _lib.ptr(idx), N, H, W,
                                      C, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1], _lib.stream_ptr(x.device))
        _lib.check(rc, "gpool_fwd")
        ctx.save_for_backward(idx)
        ctx.geom = (mode, N, C, H, W, OH, OW, k, s, p)
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        mode, N, C, H, W, OH, OW, k, s, p = ctx.geom
        gy = _nhwc(gy)
        gx = torch.empty((N, C, H, W), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
        rc = _lib.kernels().dtf_gpool(1, mode, _lib.dtype_code(gy.dtype), _lib.ptr(gy), _lib.ptr(gx), _lib.ptr(idx), N, H,
                                      W, C, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1], _lib.stream_ptr(gy.device))
        _lib.check(rc, "gpool_bwd")
        return gx, None, None, None, None


def max_pool2d(x, kernel_size, stride=None, padding=0) -> torch.Tensor:
    k = _pair(kernel_size)
    return _PoolFn.apply(x, 0, k, _pair(stride) if stride is not None else k, _pair(padding))


def avg_pool2d(x, kernel_size, stride=None, padding=0) -> torch.Tensor:
    k = _pair(kernel_size)
    return _PoolFn.apply(x, 1, k, _pair(stride) if stride is not None else k, _pair(padding))


class _GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _nhwc(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C), dtype=x.dtype, device=x.device)
        rc = _lib.kernels().dtf_gap_any(0, _lib.dtype_code(x.dtype), _lib.ptr(x), _lib.ptr(y), N, H * W, C,
                                        _lib.stream_ptr(x.device))
        _lib.check(rc, "gap_any_fwd")
        ctx.geom = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.geom
        gy = gy.contiguous()
        gx = torch.empty((N, C, H, W), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
        rc = _lib.kernels().dtf_gap_any(1, _lib.dtype_code(gy.dtype), _lib.ptr(gy), _lib.ptr(gx), N, H * W, C,
                                        _lib.stream_ptr(gy.device))
        _lib.check(rc, "gap_any_bwd")
        return gx


def global_avg_pool(x: torch.Tensor) -> torch.Tensor:
    return _GapFn.apply(x)
