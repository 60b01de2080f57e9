"""Fused optimizer apply over flat arenas.

``adam_apply`` / ``sgdm_apply`` update a contiguous fp32 parameter arena (and
its moment arenas) in one launch. They replace the per-variable
``optimizer.apply_gradients`` of the reference parameter server (reference
centralized/server.py:92,100; Keras ``Adam()`` defaults lr 1e-3, beta 0.9 /
0.999, epsilon 1e-7 come from server.py:52-55).

Hyper-parameters live in a 16-float device tensor (:class:`OptHyper`) that the
kernels read at run time, so a captured graph sees learning-rate changes and
the advancing step.

CPU tensors take the same formulas in plain torch: that is the path of the
CPU plumbing mode, not a fallback for GPU tensors.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

# formula selectors (bit 0) and modifiers (bit 1)
ADAM_KERAS, ADAM_TORCH, ADAM_DECOUPLED_WD = 0, 1, 2
SGD_KERAS, SGD_TORCH, SGD_NESTEROV = 0, 1, 2

_IDX = dict(lr=0, beta1=1, beta2=2, eps=3, weight_decay=4, grad_scale=5, momentum=6, step=7, bc1=8, bc2=9)


class OptHyper:
    """Device-resident hyper-parameter block shared by the optimizer kernels."""

    def __init__(self, device, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7, weight_decay=0.0,
                 grad_scale=1.0, momentum=0.0):
        vals = torch.zeros(16, dtype=torch.float32)
        vals[0], vals[1], vals[2], vals[3] = lr, beta1, beta2, eps
        vals[4], vals[5], vals[6] = weight_decay, grad_scale, momentum
        vals[8], vals[9] = 1.0, 1.0
        self.buf = vals.to(device)

    def set(self, **kw) -> None:
        for k, v in kw.items():
            self.buf[_IDX[k]] = float(v)

    def get(self, name: str) -> float:
        return float(self.buf[_IDX[name]])

    def tick(self) -> None:
        """step += 1 and refresh the Adam bias corrections, on the device."""
        if self.buf.is_cuda:
            _lib.check(_lib.kernels().dtf_opt_tick(_lib.ptr(self.buf), _lib.stream_ptr(self.buf.device)), "opt_tick")
        else:
            b = self.buf
            b[7] += 1.0
            b[8] = 1.0 - float(b[1]) ** float(b[7])
            b[9] = 1.0 - float(b[2]) ** float(b[7])

    def state_dict(self):
        return {"hyper": self.buf.detach().cpu().clone()}

    def load_state_dict(self, sd) -> None:
        self.buf.copy_(sd["hyper"].to(self.buf.device))


def _check_arena(name: str, t: torch.Tensor, n: int, dtype=torch.float32) -> None:
    if t.dtype != dtype or not t.is_contiguous() or t.numel() < n:
        raise ValueError(f"{name}: expected contiguous {dtype} with >= {n} elements, got {t.dtype} {tuple(t.shape)}")
    if t.data_ptr() % 16:
        raise ValueError(f"{name}: arena must be 16-byte aligned")


def adam_apply(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, hyper: OptHyper,
               shadow: torch.Tensor | None = None, mode: int = ADAM_KERAS, n: int | None = None) -> None:
    """In-place Adam step on flat arenas. ``g`` is fp32 or bf16; ``shadow`` is an
    optional bf16 copy of ``p`` refreshed in the same pass. Call ``hyper.tick()``
    once per optimizer step before the first apply of that step."""
    n = p.numel() if n is None else n
    if p.is_cuda:
        _check_arena("p", p, n); _check_arena("m", m, n); _check_arena("v", v, n)
        _check_arena("g", g, n, g.dtype)
        if shadow is not None:
            _check_arena("shadow", shadow, n, torch.bfloat16)
        rc = _lib.kernels().dtf_adam_apply(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), _lib.ptr(shadow),
                                           _lib.ptr(hyper.buf), ctypes.c_longlong(n), _lib.dtype_code(g.dtype),
                                           mode, _lib.stream_ptr(p.device))
        _lib.check(rc, "adam_apply")
        return
    h = hyper.buf
    lr, b1, b2, eps, wd, gs = (float(h[i]) for i in range(6))
    bc1, bc2 = float(h[8]), float(h[9])
    pp, mm, vv = p.view(-1)[:n], m.view(-1)[:n], v.view(-1)[:n]
    gg = g.view(-1)[:n].float() * gs
    decoupled = bool(mode & ADAM_DECOUPLED_WD)
    if not decoupled and wd:
        gg = gg + wd * pp
    mm.mul_(b1).add_(gg, alpha=1 - b1)
    vv.mul_(b2).addcmul_(gg, gg, value=1 - b2)
    if mode & ADAM_TORCH:
        upd = (lr / bc1) * mm / (vv.sqrt() / (bc2 ** 0.5) + eps)
    else:
        upd = (lr * (bc2 ** 0.5) / bc1) * mm / (vv.sqrt() + eps)
    if decoupled and wd:
        upd = upd + lr * wd * pp
    pp.sub_(upd)
    if shadow is not None:
        shadow.view(-1)[:n].copy_(pp)


def sgdm_apply(p: torch.Tensor, g: torch.Tensor, vel: torch.Tensor, hyper: OptHyper,
               shadow: torch.Tensor | None = None, mode: int = SGD_KERAS, n: int | None = None) -> None:
    """In-place SGD-with-momentum step on flat arenas (Keras or torch formula)."""
    n = p.numel() if n is None else n
    if p.is_cuda:
        _check_arena("p", p, n); _check_arena("vel", vel, n); _check_arena("g", g, n, g.dtype)
        if shadow is not None:
            _check_arena("shadow", shadow, n, torch.bfloat16)
        rc = _lib.kernels().dtf_sgdm_apply(_lib.ptr(p), _lib.ptr(g), _lib.ptr(vel), _lib.ptr(shadow),
                                           _lib.ptr(hyper.buf), ctypes.c_longlong(n), _lib.dtype_code(g.dtype),
                                           mode, _lib.stream_ptr(p.device))
        _lib.check(rc, "sgdm_apply")
        return
    h = hyper.buf
    lr, wd, gs, mu = float(h[0]), float(h[4]), float(h[5]), float(h[6])
    pp, vv = p.view(-1)[:n], vel.view(-1)[:n]
    gg = g.view(-1)[:n].float() * gs + wd * pp
    nesterov = bool(mode & SGD_NESTEROV)
    if mode & SGD_TORCH:
        vv.mul_(mu).add_(gg)
        pp.sub_(lr * (gg + mu * vv if nesterov else vv))
    else:
        vv.mul_(mu).sub_(lr * gg)
        pp.add_(mu * vv - lr * gg if nesterov else vv)
    if shadow is not None:
        shadow.view(-1)[:n].copy_(pp)
