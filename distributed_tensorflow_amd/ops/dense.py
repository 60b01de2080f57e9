"""Dense (fully connected) layer op: ``y = act(x @ W^T + b)``.

On a GPU with bf16 operands the forward, the data gradient and the weight
gradient run on the hand-written MFMA GEMM (``ops/gemm.py``) with bias and
ReLU/GELU applied in the epilogue, and the bias gradient is a fused column
reduction; fp32 operands use the exact f32 library GEMM. On CPU the same maths
runs in plain torch. This is the MatMul + BiasAdd + ReLU chain of the
reference model (reference initializer.py:16-18).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import gemm as _gemm


def _act_fwd(y, activation):
    if activation == "relu":
        return F.relu(y)
    if activation == "gelu":
        return F.gelu(y)
    return y


class _LinearFn(torch.autograd.Function):
    """bf16 GPU path: every GEMM on the MFMA kernel, epilogues fused."""

    @staticmethod
    def forward(ctx, x, w, b, activation):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if activation == "gelu":
            saved = _gemm.gemm_nt(x2, w, bias=b, activation=None)          # pre-activation
            y = _gemm.gelu_fwd(saved)
        else:
            y = _gemm.gemm_nt(x2, w, bias=b, activation=activation)        # [M, N]
            saved = y if activation == "relu" else None
        ctx.activation = activation
        ctx.bias_dtype = b.dtype if b is not None else None
        ctx.x_shape = x.shape
        ctx.save_for_backward(x2, w, saved)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w, saved = ctx.saved_tensors
        g = gy.reshape(-1, gy.shape[-1])
        if ctx.activation == "relu":
            g = _gemm.relu_bwd(g, saved)
        elif ctx.activation == "gelu":
            g = _gemm.gelu_bwd(g, saved)
        g = g.contiguous()
        gx = _gemm.gemm_nn(g, w) if ctx.needs_input_grad[0] else None     # [M, K]
        gw = _gemm.gemm_tn(g, x2) if ctx.needs_input_grad[1] else None    # [N, K]
        gb = None
        if ctx.bias_dtype is not None and ctx.needs_input_grad[2]:
            gb = _gemm.col_sum(g, dtype=ctx.bias_dtype)
        if gx is not None:
            gx = gx.reshape(ctx.x_shape)
        return gx, gw, gb, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None,
           activation: str | None = None) -> torch.Tensor:
    """``act(x @ weight.T + bias)`` with ``weight`` in torch ``(out, in)`` layout."""
    if x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and _gemm.supported(x, weight):
        return _LinearFn.apply(x, weight, bias, activation)
    if x.dtype != weight.dtype:
        x = x.to(weight.dtype)
    y = F.linear(x, weight, bias.to(weight.dtype) if bias is not None else None)
    return _act_fwd(y, activation)
