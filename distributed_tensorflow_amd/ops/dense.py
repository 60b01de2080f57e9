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
from . import rng as _rng


def _act_fwd(y, activation):
    if activation == "relu":
        return F.relu(y)
    if activation == "gelu":
        return F.gelu(y)
    return y


class _LinearFn(torch.autograd.Function):
    """bf16 GPU path: every GEMM on the MFMA kernel, epilogues fused."""

    @staticmethod
    def forward(ctx, x, w, b, activation, drop_p=0.0):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        ctx.drop = None
        drop = None
        if drop_p > 0.0:
            st = _rng.state_for(x.device)
            ctx.drop = (drop_p, st.seed, st.snapshot())
            drop = ctx.drop
        if activation == "gelu":
            saved = _gemm.gemm_nt(x2, w, bias=b, activation=None)          # pre-activation
            y = _gemm.gelu_fwd(saved)
            if drop is not None:
                y = _gemm.dropout_fwd(y, *drop)
        else:
            y = _gemm.gemm_nt(x2, w, bias=b, activation=activation, dropout=drop)   # [M, N]
            saved = y if activation == "relu" else None
        if drop is not None:
            st.advance(y.numel())
        ctx.activation = activation
        ctx.bias_dtype = b.dtype if b is not None else None
        ctx.x_shape = x.shape
        ctx.save_for_backward(x2, w, saved)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy, g_alias=None):
        """``g_alias`` (fork variant): gradient that arrived through the aliased input; added to the data
        gradient inside the GEMM epilogue."""
        x2, w, saved = ctx.saved_tensors
        g = gy.reshape(-1, gy.shape[-1])
        if ctx.drop is not None:
            g = _gemm.dropout_bwd(g, *ctx.drop, activation=ctx.activation, saved=saved)
        elif ctx.activation == "relu":
            g = _gemm.relu_bwd(g, saved)
        elif ctx.activation == "gelu":
            g = _gemm.gelu_bwd(g, saved)
        g = g.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            add = None
            if g_alias is not None and g_alias.dtype == torch.bfloat16 and g_alias.is_contiguous() \
                    and g_alias.data_ptr() % 16 == 0:
                add, g_alias = g_alias.reshape(-1, g_alias.shape[-1]), None
            gx = _gemm.gemm_nn(g, w, add=add)                               # [M, K]
        gw = _gemm.gemm_tn(g, x2) if ctx.needs_input_grad[1] else None    # [N, K]
        gb = None
        if ctx.bias_dtype is not None and ctx.needs_input_grad[2]:
            gb = _gemm.col_sum(g, dtype=ctx.bias_dtype)
        if gx is not None:
            gx = gx.reshape(ctx.x_shape)
            if g_alias is not None:
                gx = gx + g_alias
        return gx, gw, gb, None, None


class _LinearForkFn(_LinearFn):
    """``(linear(x), alias of x)``: when ``x`` also feeds a residual branch, routing that use through the
    alias hands its gradient to this backward, which accumulates it in the data-gradient GEMM's
    epilogue instead of leaving the sum to a separate add kernel."""

    @staticmethod
    def forward(ctx, x, w, b, activation, drop_p=0.0):
        return _LinearFn.forward(ctx, x, w, b, activation, drop_p), x.view_as(x)

    @staticmethod
    def backward(ctx, gy, g_alias):
        if gy is None:
            return g_alias, None, None, None, None
        return _LinearFn.backward(ctx, gy, g_alias)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None,
           activation: str | None = None, dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    """``dropout(act(x @ weight.T + bias))`` with ``weight`` in torch ``(out, in)`` layout. On the bf16
    GPU path the dropout mask comes from the Philox state of ``ops/rng.py`` and is applied in the
    GEMM epilogue; elsewhere ``torch.nn.functional.dropout`` is used."""
    p = float(dropout_p) if training else 0.0
    if x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and _gemm.supported(x, weight):
        return _LinearFn.apply(x, weight, bias, activation, p)
    if x.dtype != weight.dtype:
        x = x.to(weight.dtype)
    y = F.linear(x, weight, bias.to(weight.dtype) if bias is not None else None)
    y = _act_fwd(y, activation)
    return F.dropout(y, p, True) if p > 0.0 else y


def linear_fork(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None,
                activation: str | None = None):
    """``(linear(x, ...), x')`` with ``x'`` an alias of ``x`` for its other consumer (a residual branch)."""
    if (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and _gemm.supported(x, weight)
            and x.requires_grad and torch.is_grad_enabled()):
        return _LinearForkFn.apply(x, weight, bias, activation, 0.0)
    return linear(x, weight, bias, activation), x
