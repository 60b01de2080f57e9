"""Fused softmax + sparse categorical cross-entropy.

``softmax_cross_entropy(logits, labels)`` returns the per-example loss vector
with no reduction, which is what the reference's ``loss`` helper returns
(reference centralized/client.py:11-13); ``reduction='sum'`` reproduces the
gradient the reference's ``tape.gradient`` of that vector computes
(client.py:16-19), ``'mean'`` is the usual default elsewhere.

``probs_clip_eps`` selects the reference's exact formulation, which applies the
loss to softmax *probabilities* clipped to ``[eps, 1 - eps]`` and renormalised
(what Keras does for ``from_logits=False``); it runs in plain torch since it is
a parity option for 10-class outputs, not a hot path.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _label_code(y: torch.Tensor) -> int:
    if y.dtype == torch.int32:
        return 0
    if y.dtype == torch.int64:
        return 1
    raise TypeError("labels must be int32 or int64")


def _row_padded(t: torch.Tensor) -> bool:
    """Rows that are read in place: contiguous, or padded to the next multiple of 16 bytes in a 16-byte aligned
    buffer (the layout ``ops.dense.linear`` gives logits whose width is not a multiple of 8). Any other view -- a
    column slice, rows of a much wider matrix -- is copied: backward allocates its gradient with the same row
    stride and needs the same alignment as the input."""
    if t.dim() != 2 or t.stride(1) != 1 or t.shape[0] <= 0:
        return False
    C, ld = t.shape[1], t.stride(0)
    if ld == C:
        return True
    per16 = 16 // t.element_size()
    return t.data_ptr() % 16 == 0 and ld == -(-C // per16) * per16


def _fwd(logits: torch.Tensor, labels: torch.Tensor, want_correct: bool):
    B, C = logits.shape
    if logits.is_cuda:
        # rows may be padded (stride(0) > C): the logits of a vocabulary whose size is not a multiple of 8 come in a
        # buffer with 16-byte aligned rows (ops.dense.linear), and are read in place
        x = logits if _row_padded(logits) else logits.contiguous()
        y = labels.contiguous()
        loss = torch.empty(B, dtype=torch.float32, device=x.device)
        lse = torch.empty(B, dtype=torch.float32, device=x.device)
        correct = torch.empty(B, dtype=torch.uint8, device=x.device) if want_correct else None
        rc = _lib.kernels().dtf_softmax_ce_fwd_ld(_lib.ptr(x), _lib.ptr(y), _lib.ptr(loss), _lib.ptr(lse),
                                                  _lib.ptr(correct), B, C, ctypes.c_longlong(x.stride(0)),
                                                  _lib.dtype_code(x.dtype), _label_code(y), _lib.stream_ptr(x.device))
        _lib.check(rc, "softmax_ce_fwd")
        return x, y, loss, lse, correct
    xf = logits.float()
    lse = torch.logsumexp(xf, dim=1)
    valid = (labels >= 0) & (labels < C)
    safe = labels.clamp(0, C - 1).long()
    loss = torch.where(valid, lse - xf.gather(1, safe[:, None])[:, 0], torch.zeros_like(lse))
    correct = (xf.argmax(1) == labels).to(torch.uint8) if want_correct else None
    return logits, labels, loss, lse, correct


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        x, y, loss, lse, _ = _fwd(logits, labels, False)
        ctx.save_for_backward(x, y, lse)
        return loss

    @staticmethod
    def backward(ctx, gout):
        x, y, lse = ctx.saved_tensors
        B, C = x.shape
        g = gout.contiguous().float()
        if x.is_cuda:
            ld = x.stride(0)
            # same row padding as the logits; the kernel zeroes the padding of the gradient
            dx = torch.empty((B, ld), dtype=x.dtype, device=x.device)[:, :C]
            rc = _lib.kernels().dtf_softmax_ce_bwd_ld(_lib.ptr(x), _lib.ptr(y), _lib.ptr(lse), _lib.ptr(g), _lib.ptr(dx),
                                                      B, C, ctypes.c_longlong(ld), _lib.dtype_code(x.dtype),
                                                      _label_code(y), _lib.stream_ptr(x.device))
            _lib.check(rc, "softmax_ce_bwd")
            return dx, None
        valid = ((y >= 0) & (y < C))
        p = torch.exp(x.float() - lse[:, None])
        p[torch.arange(B), y.clamp(0, C - 1).long()] -= 1.0
        dx = p * (g * valid)[:, None]
        return dx.to(x.dtype), None


def _clipped_probs_loss(logits, labels, eps):
    p = torch.softmax(logits.float(), dim=1).clamp(eps, 1.0 - eps)
    q = p / p.sum(dim=1, keepdim=True)
    return -torch.log(q.gather(1, labels.long()[:, None])[:, 0])


def softmax_cross_entropy(logits: torch.Tensor, labels: torch.Tensor, reduction: str = "none",
                          probs_clip_eps: float | None = None) -> torch.Tensor:
    """Loss of ``[B, C]`` logits against ``[B]`` integer labels (label < 0: ignored row)."""
    if logits.dim() != 2:
        logits = logits.reshape(-1, logits.shape[-1])
        labels = labels.reshape(-1)
    if probs_clip_eps is not None:
        loss = _clipped_probs_loss(logits, labels, probs_clip_eps)
    else:
        loss = _SoftmaxCE.apply(logits, labels)
    if reduction == "none":
        return loss
    if reduction == "sum":
        return loss.sum()
    if reduction == "mean":
        n = (labels >= 0).sum().clamp(min=1)
        return loss.sum() / n
    raise ValueError(f"unknown reduction {reduction!r}")


@torch.no_grad()
def softmax_ce_eval(logits: torch.Tensor, labels: torch.Tensor):
    """(per-example loss, correct flags) in one pass; used by evaluation loops
    (reference centralized/server.py:24-37 computes argmax accuracy separately)."""
    if logits.dim() != 2:
        logits = logits.reshape(-1, logits.shape[-1])
        labels = labels.reshape(-1)
    _, _, loss, _, correct = _fwd(logits, labels, True)
    return loss, correct
