"""Multi-head self-attention on the batched MFMA GEMM plus the fused softmax kernels.

Input is the fused QKV projection ``qkv[B, S, 3, nh, hd]`` (one GEMM output); the three products
``S = Q K^T``, ``O = P V`` and their four gradient products address Q, K and V *inside* that
buffer through leading dimensions and two-level batch strides (batch, head), so no head
split / transpose copies are made. The scale ``1/sqrt(hd)`` and the optional additive
key-padding bias are fused into the softmax kernel.

Two paths: head dimension 64 with ``S % 128 == 0`` (the BERT-base configuration) runs the fused
flash-style kernels of ``csrc/kernels/flash_attn.hip`` (scores never leave registers; backward
recomputes the probabilities from the stored log-sum-exp). Other shapes materialise the scores
(``[B, nh, S, S]`` bf16) and run the batched GEMM + fused softmax kernels. ``set_flash(False)`` or
``DTF_FLASH=0`` forces the second path.
"""
from __future__ import annotations

import ctypes
import math
import os

import torch

from . import _lib
from . import gemm as G


_flash = os.environ.get("DTF_FLASH", "1") == "1"


def set_flash(on: bool) -> None:
    global _flash
    _flash = bool(on)


def _flash_ok(qkv: torch.Tensor) -> bool:
    B, S, three, nh, hd = qkv.shape
    return _flash and hd == 64 and S % 128 == 0 and S >= 128


class _FlashFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias):
        B, S, _, nh, hd = qkv.shape
        out = torch.empty(B, S, nh, hd, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, nh, S, dtype=torch.float32, device=qkv.device)
        rc = _lib.kernels().dtf_flash_attn_fwd(_lib.ptr(qkv), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(lse), B, S, nh, hd,
                                               ctypes.c_float(1.0 / math.sqrt(hd)), _lib.stream_ptr(qkv.device))
        _lib.check(rc, "flash_attn_fwd")
        ctx.save_for_backward(qkv, bias if bias is not None else lse, out, lse)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, bias, out, lse = ctx.saved_tensors
        B, S, _, nh, hd = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, nh, S, dtype=torch.float32, device=qkv.device)
        # the kernels also sum the columns of the dqkv they store (per wave, finished by one small kernel): the bias
        # gradient of the fused QKV projection, offered to that layer's backward under dqkv's address
        from . import layernorm as _ln
        parts = sums = None
        if _ln.hand_over_enabled():
            parts = torch.empty((B * (S // 32), 3 * nh * hd), dtype=torch.float32, device=qkv.device)
            sums = torch.empty(3 * nh * hd, dtype=torch.float32, device=qkv.device)
        rc = _lib.kernels().dtf_flash_attn_bwd_cs(_lib.ptr(qkv), _lib.ptr(bias if ctx.has_bias else None), _lib.ptr(out),
                                                  _lib.ptr(lse), _lib.ptr(dout), _lib.ptr(delta), _lib.ptr(dqkv), B, S,
                                                  nh, hd, ctypes.c_float(1.0 / math.sqrt(hd)), _lib.ptr(parts),
                                                  _lib.ptr(sums), _lib.stream_ptr(qkv.device))
        _lib.check(rc, "flash_attn_bwd")
        if sums is not None:
            _ln.offer_col_sum(dqkv, sums)
        return dqkv, None


def _fast(qkv: torch.Tensor) -> bool:
    B, S, three, nh, hd = qkv.shape
    return (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.is_contiguous() and hd % 8 == 0 and S % 8 == 0
            and S <= 4096 and three == 3)


def _softmax_fwd(scores, bias, rows_per_batch, scale):
    L = scores.shape[-1]
    rows = scores.numel() // L
    rc = _lib.kernels().dtf_softmax_fwd(_lib.ptr(scores), _lib.ptr(scores), _lib.ptr(bias), ctypes.c_longlong(rows), L,
                                        ctypes.c_longlong(rows_per_batch), ctypes.c_float(scale),
                                        _lib.stream_ptr(scores.device))
    _lib.check(rc, "softmax_fwd")
    return scores


def _softmax_bwd(p, dp, scale):
    L = p.shape[-1]
    rows = p.numel() // L
    rc = _lib.kernels().dtf_softmax_bwd(_lib.ptr(p), _lib.ptr(dp), _lib.ptr(dp), ctypes.c_longlong(rows), L,
                                        ctypes.c_float(scale), _lib.stream_ptr(p.device))
    _lib.check(rc, "softmax_bwd")
    return dp


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bias):
        B, S, _, nh, hd = qkv.shape
        dev = qkv.device
        ld = 3 * nh * hd                      # row stride of a token inside qkv
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]   # views; only their data_ptr is used
        scores = torch.empty(B, nh, S, S, dtype=qkv.dtype, device=dev)
        sb = (S * ld, hd)                     # (batch, head) strides inside qkv
        sc = (nh * S * S, S * S)
        G.gemm_raw(G.NT, q, k, scores, S, S, hd, ld, ld, S, extra=G.batch_dims(B, nh, sb, sb, sc))
        p = _softmax_fwd(scores, bias, nh * S, 1.0 / math.sqrt(hd))
        out = torch.empty(B, S, nh, hd, dtype=qkv.dtype, device=dev)
        so = (S * nh * hd, hd)
        G.gemm_raw(G.NN, p, v, out, S, hd, S, S, ld, nh * hd, extra=G.batch_dims(B, nh, sc, sb, so))
        ctx.save_for_backward(qkv, p)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, p = ctx.saved_tensors
        B, S, _, nh, hd = qkv.shape
        dev = qkv.device
        ld = 3 * nh * hd
        dout = dout.contiguous()
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]
        sb = (S * ld, hd)
        sc = (nh * S * S, S * S)
        so = (S * nh * hd, hd)
        lo = nh * hd
        # dV = P^T dO
        G.gemm_raw(G.TN, p, dout, dv, S, hd, S, S, lo, ld, extra=G.batch_dims(B, nh, sc, so, sb))
        # dP = dO V^T
        dp = torch.empty_like(p)
        G.gemm_raw(G.NT, dout, v, dp, S, S, hd, lo, ld, S, extra=G.batch_dims(B, nh, so, sb, sc))
        ds = _softmax_bwd(p, dp, 1.0 / math.sqrt(hd))
        # dQ = dS K ; dK = dS^T Q
        G.gemm_raw(G.NN, ds, k, dq, S, hd, S, S, ld, ld, extra=G.batch_dims(B, nh, sc, sb, sb))
        G.gemm_raw(G.TN, ds, q, dk, S, hd, S, S, ld, ld, extra=G.batch_dims(B, nh, sc, sb, sb))
        return dqkv, None


def self_attention(qkv: torch.Tensor, key_bias: torch.Tensor | None = None) -> torch.Tensor:
    """``qkv[B, S, 3, nh, hd]`` -> context ``[B, S, nh, hd]``. ``key_bias[B, S]`` (fp32) is added to
    the scores of every query of that batch element (0 for real tokens, -inf for padding)."""
    if _fast(qkv):
        if key_bias is not None:
            key_bias = key_bias.float().contiguous()
        if _flash_ok(qkv):
            return _FlashFn.apply(qkv, key_bias)
        return _AttnFn.apply(qkv, key_bias)
    B, S, _, nh, hd = qkv.shape
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).float() for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :].float()
    o = torch.softmax(s, dim=-1) @ v
    return o.permute(0, 2, 1, 3).to(qkv.dtype).contiguous()
