"""LayerNorm with the residual add fused in (front end of ``csrc/kernels/transformer.hip``).

``layer_norm(x, gamma, beta, eps, residual=None, return_sum=False)`` normalises ``x (+ residual)``
over the last axis. With ``return_sum=True`` it also returns the bf16 sum, the new residual
stream, produced by the same kernel pass. bf16 activations on a GPU use the HIP kernels;
everything else uses the torch composition.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F

from . import _lib

_ws: dict = {}


def _scratch(device, numel):
    key = (device.type, device.index)
    t = _ws.get(key)
    if t is None or t.numel() < numel:
        t = torch.empty(max(numel, 1 << 20), dtype=torch.float32, device=device)
        _ws[key] = t
    return t


# Hand-over to the linear layer whose output a norm normalised: the norm's backward kernel also sums the columns of
# the gradient dx it writes -- that layer's bias gradient -- and leaves them here under dx's address. The entry keeps
# dx alive (its address cannot be reused) and its version counter; the taker checks both, so a gradient that autograd
# summed with another one (new tensor, or in place: new version) never matches. At most two entries are kept.
_handed: dict = {}
_hand_over = __import__("os").environ.get("DTF_LN_BIAS_GRAD", "1") != "0"


def hand_over_enabled() -> bool:
    return _hand_over


def offer_col_sum(t: torch.Tensor, cs: torch.Tensor) -> None:
    """Leaves the fp32 column sums ``cs`` of ``t`` (viewed as ``[rows, cs.numel()]``) for the layer whose output
    gradient ``t`` is (another producer than the LayerNorm backward: the fused attention's backward kernels)."""
    while len(_handed) >= 2:
        _handed.pop(next(iter(_handed)))
    _handed[t.data_ptr()] = (t, t._version, cs)


def take_col_sum(g: torch.Tensor):
    """fp32 column sums of ``g`` (viewed as ``[rows, H]``) when ``g`` is, untouched, the input gradient the last
    LayerNorm backward produced; ``None`` otherwise."""
    ent = _handed.pop(g.data_ptr(), None) if _handed else None
    if ent is None:
        return None
    dx, version, cs = ent
    if (g._version != version or g.numel() != dx.numel() or g.shape[-1] != cs.numel() or g.dtype != dx.dtype
            or not g.is_contiguous()):
        return None
    return cs


def _fast(x, gamma):
    H = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.bfloat16 and H % 8 == 0 and H <= 2048 and x.is_contiguous()
            and gamma.dtype in (torch.float32, torch.bfloat16))


class _LNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, want_sum):
        H = x.shape[-1]
        rows = x.numel() // H
        dev = x.device
        y = torch.empty_like(x)
        s = torch.empty_like(x) if res is not None else None   # x + res, written by the kernel: backward's input
        stats = torch.empty(rows, 2, dtype=torch.float32, device=dev)
        if res is not None:
            res = res.contiguous()
        rc = _lib.kernels().dtf_ln_fwd(_lib.ptr(x), _lib.ptr(res), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y),
                                       _lib.ptr(s), _lib.ptr(stats), ctypes.c_longlong(rows), H, ctypes.c_float(eps),
                                       _lib.dtype_code(gamma.dtype), _lib.stream_ptr(dev))
        _lib.check(rc, "ln_fwd")
        xin = x if res is None else s
        ctx.save_for_backward(xin, gamma, stats)
        ctx.has_res = res is not None
        ctx.want_sum = want_sum
        if want_sum:
            return y, (s if s is not None else x.view_as(x))
        return y, None

    @staticmethod
    def backward(ctx, dy, dsum):
        xin, gamma, stats = ctx.saved_tensors
        H = xin.shape[-1]
        rows = xin.numel() // H
        dev = xin.device
        dy = dy.contiguous()
        dres_in = dsum.contiguous() if dsum is not None else None
        dx = torch.empty_like(xin)
        lib = _lib.kernels()
        nb = lib.dtf_ln_bwd_num_partials(ctypes.c_longlong(rows))
        part = _scratch(dev, nb * 3 * H)
        dgb = torch.empty(3, H, dtype=torch.float32, device=dev)
        rc = lib.dtf_ln_bwd(_lib.ptr(xin), _lib.ptr(dy), _lib.ptr(dres_in), _lib.ptr(gamma), _lib.ptr(stats), _lib.ptr(dx),
                            _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(part), ctypes.c_longlong(rows), H,
                            _lib.dtype_code(gamma.dtype), _lib.ptr(dgb[2]) if _hand_over else None,
                            _lib.stream_ptr(dev))
        _lib.check(rc, "ln_bwd")
        if _hand_over:
            while len(_handed) >= 2:
                _handed.pop(next(iter(_handed)))
            _handed[dx.data_ptr()] = (dx, dx._version, dgb[2])
        return dx, (dx if ctx.has_res else None), dgb[0].to(gamma.dtype), dgb[1].to(gamma.dtype), None, None


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-12,
               residual: torch.Tensor | None = None, return_sum: bool = False):
    """``LN(x + residual)``; returns ``y`` or ``(y, x + residual)``."""
    if _fast(x, gamma):
        y, s = _LNFn.apply(x, residual, gamma, beta, eps, return_sum)
        return (y, s) if return_sum else y
    s = x if residual is None else x + residual
    y = F.layer_norm(s.float(), (s.shape[-1],), gamma.float(), beta.float(), eps).to(x.dtype)
    return (y, s) if return_sum else y
