"""Batch normalisation fused with residual add and ReLU (front end of ``csrc/kernels/norm.hip``).

``batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu,
residual)`` computes ``act(bn(x) + residual)`` for a channels_last bf16 activation in two
memory passes forward and two backward. fp32 activations, channel counts that are not a
multiple of 8 and CPU tensors use the equivalent torch composition.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F

from . import _lib

_ws: dict = {}
_pool_fused = __import__("os").environ.get("DTF_BN_POOL_FUSED", "1") != "0"


def _scratch(device, numel: int) -> torch.Tensor:
    key = (device.type, device.index)
    t = _ws.get(key)
    if t is None or t.numel() < numel:
        t = torch.empty(max(numel, 1 << 21), dtype=torch.float32, device=device)
        _ws[key] = t
    return t


# Every norm with a ReLU writes mask bits in its forward pass (form 3 of the backward kernels), also those without a
# residual, whose mask the backward pass could recompute from the input (form 2): one byte per 8 elements more
# traffic, but the backward kernels then need no scale / shift registers (66 against 89, seven waves per SIMD against
# five): 13,964 against 13,906 images/s. ``DTF_BN_MASK_ALL=0`` restores the recomputed mask.
_mask_all = __import__("os").environ.get("DTF_BN_MASK_ALL", "1") != "0"
_counters: dict = {}
# Off by default: measured on MI355X (ResNet-50, batch 1024) the single launch takes 13.7-15.2 us against 8.8 + 5.5 us
# for the fold and finalize kernels it replaces (80 launches fewer per step, 13,640 vs 13,700 images/s): the device-
# scope store, counter and re-read of the folded rows are three dependent round trips to the memory side of the
# fabric, as long as the second launch they save.
_fused_reduce = __import__("os").environ.get("DTF_BN_FUSED_REDUCE", "0") == "1"


def _reduce_counters(device, C: int):
    """Zeroed counters for one launch of the fused fold + finalize kernel (``bn_reduce_finalize_kernel`` leaves them
    at zero again). Consecutive launches take consecutive slots of a ring, so launches that overlap on different
    streams do not share counters. ``None`` (the default, see ``_fused_reduce``) selects the separate fold and
    finalize kernels."""
    if not _fused_reduce:
        return None
    key = (device.type, device.index)
    ent = _counters.get(key)
    if ent is None:
        ent = _counters[key] = [torch.zeros(8192, dtype=torch.int32, device=device), 0]
    need = _lib.kernels().dtf_bn_reduce_counters(C)
    buf, pos = ent
    if pos + need > buf.numel():
        pos = 0
    ent[1] = pos + need
    return ctypes.c_void_p(buf.data_ptr() + 4 * pos)


def _fast_path(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last))


class GradStatsRequest:
    """Hand-over between a batch norm and the convolution that consumes its output. The norm's forward fills in
    what its backward reductions need (input, statistics, ReLU mask); the convolution's backward, whose
    data-gradient GEMM produces the very gradient those reductions run over, performs them while it stores that
    gradient and leaves ``partials`` here together with the gradient they belong to. The norm's
    backward uses them only if the gradient it receives is that tensor, unchanged: exactly one convolution picked
    the request up (``consumers``), the address matches, and the tensor's version counter has not moved -- autograd
    sums the gradients of several consumers either into a new tensor (another address) or in place (another
    version); the request keeps the tensor alive meanwhile, so its address cannot be reused. Otherwise the norm
    reduces on its own."""
    __slots__ = ("x", "stats", "mask", "relu", "partials", "dy_ptr", "dy_ref", "dy_version", "consumers")

    def __init__(self):
        self.x = self.stats = self.mask = self.partials = self.dy_ref = None
        self.relu, self.dy_ptr, self.dy_version, self.consumers = 0, 0, -1, 0

    def hand_over(self, partials, dy: torch.Tensor) -> None:
        self.partials, self.dy_ref, self.dy_ptr, self.dy_version = partials, dy, dy.data_ptr(), dy._version

    def matches(self, dy: torch.Tensor) -> bool:
        return (self.partials is not None and self.consumers == 1 and self.dy_ptr == dy.data_ptr()
                and self.dy_version == dy._version and self.dy_ref is not None and dy.shape == self.dy_ref.shape)


class ForkToken:
    """Marks the alias output of ``ops.conv.conv2d_fork``. When that alias is the residual input of a
    norm-add-ReLU, the norm's backward does not write the masked gradient of the residual branch: it hands the
    incoming gradient itself to the branch and leaves the ReLU mask here (``masked_grad`` = that gradient, its
    version counter, mask bytes); the fork's backward, which accumulates the branch gradient in its data-gradient GEMM
    epilogue anyway, applies the mask there. One pass over the widest activations of the network is saved.
    The output of a plain batch norm (no ReLU, no residual: the norm of a down-sampling branch) carries a token too:
    its own backward reads the gradient together with the mask bits (the kernels' ``relu == 3`` form) instead of a
    masked copy. That norm also leaves its input, weight and statistics here (``owner``): the norm-add-ReLU then runs
    the plain norm's backward inside its own (one apply pass for both inputs, the gradient and the bits read once)
    and leaves the results in ``done`` for the plain norm's backward node to return."""
    __slots__ = ("masked_grad", "owner", "done")

    def __init__(self):
        self.masked_grad = None
        self.owner = None
        self.done = None


_TOKEN_ATTR = "_dtf_fork_token"
_REQ_ATTR = "_dtf_grad_stats_request"


def request_of(t: torch.Tensor):
    """The :class:`GradStatsRequest` attached to a batch-norm output (None for any other tensor). Every call is a
    consumer picking the request up; with more than one the hand-over is off."""
    req = getattr(t, _REQ_ATTR, None)
    if req is not None:
        req.consumers += 1
    return req


class DeferredNorm:
    """A batch norm (+ ReLU) whose apply pass has not run: ``tensor`` holds the norm's INPUT (it is the autograd
    handle of the norm's OUTPUT: gradients that flow into it are gradients of the normalised activation), ``scale``
    and ``shift`` the per-channel fp32 coefficients. ``ops.conv.conv2d`` takes this object in place of an activation
    and normalises while it stages its operand, forward and in its weight gradient, so the normalised tensor is
    never written or re-read. Anything else that needs the activation calls :meth:`materialize`."""
    __slots__ = ("tensor", "scale", "shift", "relu")

    def __init__(self, tensor, scale, shift, relu):
        self.tensor, self.scale, self.shift, self.relu = tensor, scale, shift, relu

    @property
    def shape(self):
        return self.tensor.shape

    def materialize(self) -> torch.Tensor:
        y = _ApplyFn.apply(self.tensor, self.scale, self.shift, self.relu)
        req = getattr(self.tensor, _REQ_ATTR, None)
        if req is not None:
            req.consumers += 1        # the copy is a second consumer of the alias: no hand-over of backward sums
        return y


class _ApplyFn(torch.autograd.Function):
    """``relu(x * scale + shift)`` of a deferred norm as a tensor; its backward masks by the output."""

    @staticmethod
    def forward(ctx, x, scale, shift, relu):
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        rc = _lib.kernels().dtf_bn_fwd_apply(_lib.ptr(x), None, _lib.ptr(y), _lib.ptr(scale), _lib.ptr(shift),
                                             ctypes.c_longlong(N * H * W), C, int(relu), _lib.stream_ptr(x.device))
        _lib.check(rc, "bn_fwd_apply")
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, g):
        # the alias already stands for the norm's output: the gradient passes through unchanged (the norm's own
        # backward applies the ReLU mask it recomputes from its input)
        return g, None, None, None


class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, momentum, eps, relu, residual, partials=None, request=None,
                defer=False, out_token=None, res_scale=None, res_shift=None):
        """``defer``: finish the statistics only and return an alias of ``x`` (see :class:`DeferredNorm`); needs
        ``partials`` and no residual. ``out_token``: the :class:`ForkToken` of this norm's output (plain norms
        only). ``res_scale`` / ``res_shift``: ``residual`` is the alias of a plain norm whose apply pass has not run
        (:func:`batch_norm_lazy`); the apply kernel normalises it on the fly (needs ``partials``)."""
        ctx.out_token = out_token
        N, C, H, W = x.shape
        M = N * H * W
        dev = x.device
        y = None if defer else torch.empty_like(x)  # preserves channels_last
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)
        lib = _lib.kernels()
        nb = lib.dtf_bn_num_partials(ctypes.c_longlong(M), C)
        ws = _scratch(dev, nb * 2 * C)
        if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
            residual = residual.contiguous(memory_format=torch.channels_last)
        # relu mask for backward: with a residual the sign of the pre-activation cannot be recomputed
        # from x alone, so the forward pass writes it as one bit per element (1/16 of re-reading y);
        # without a residual it could be recomputed from x (see _mask_all)
        ctx.has_res = residual is not None
        # (a deferred norm runs no apply pass, so nothing would write the bits: it keeps the recomputed mask)
        ctx.relu = 0 if not relu else (3 if (residual is not None or (_mask_all and not defer)) else 2)
        ctx.fork_token = getattr(residual, _TOKEN_ATTR, None) if (residual is not None and relu) else None
        mask = torch.empty(M * C // 8, dtype=torch.uint8, device=dev) if ctx.relu == 3 else None
        if partials is not None:
            # the producer of x already reduced sum(x), sum(x^2) to partial rows: x is read once, by the apply pass
            if partials.dtype != torch.float32 or partials.dim() != 3 or tuple(partials.shape[1:]) != (2, C) \
                    or not partials.is_contiguous():
                raise ValueError("batch_norm_act: partials must be a contiguous fp32 [rows, 2, C] tensor")
            ws = _scratch(dev, max(nb, 512) * 2 * C)
            if res_scale is not None:
                rc = lib.dtf_bn_fwd_train_stats_rn(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(y), _lib.ptr(gamma),
                                                   _lib.ptr(beta), _lib.ptr(run_mean), _lib.ptr(run_var),
                                                   _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(stats[2]),
                                                   _lib.ptr(stats[3]), _lib.ptr(partials), int(partials.shape[0]),
                                                   _lib.ptr(ws), _lib.ptr(mask), ctypes.c_longlong(M), C,
                                                   ctypes.c_float(eps), ctypes.c_float(momentum), int(relu),
                                                   _reduce_counters(dev, C), _lib.ptr(res_scale), _lib.ptr(res_shift),
                                                   _lib.stream_ptr(dev))
            else:
                rc = lib.dtf_bn_fwd_train_stats(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(y), _lib.ptr(gamma),
                                                _lib.ptr(beta), _lib.ptr(run_mean), _lib.ptr(run_var), _lib.ptr(stats[0]),
                                                _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]),
                                                _lib.ptr(partials), int(partials.shape[0]), _lib.ptr(ws), _lib.ptr(mask),
                                                ctypes.c_longlong(M), C, ctypes.c_float(eps), ctypes.c_float(momentum),
                                                int(relu), _reduce_counters(dev, C), _lib.stream_ptr(dev))
        else:
            if res_scale is not None:
                raise ValueError("batch_norm_act: a lazy residual needs the partial statistics of x")
            rc = lib.dtf_bn_fwd_train(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta),
                                      _lib.ptr(run_mean), _lib.ptr(run_var), _lib.ptr(stats[0]), _lib.ptr(stats[1]),
                                      _lib.ptr(stats[2]), _lib.ptr(stats[3]), _lib.ptr(ws), _lib.ptr(mask),
                                      ctypes.c_longlong(M), C, ctypes.c_float(eps), ctypes.c_float(momentum),
                                      int(relu), _lib.stream_ptr(dev))
        _lib.check(rc, "bn_fwd_train")
        ctx.save_for_backward(x, mask, gamma, stats)
        ctx.request = request
        if out_token is not None and not relu and residual is None:
            out_token.owner = (x, gamma, stats)
        if request is not None:
            request.x, request.stats, request.mask, request.relu = x, stats, mask, ctx.relu
        if defer:
            ctx.mark_non_differentiable(stats)
            return x.view_as(x), stats
        return y

    @staticmethod
    def backward(ctx, dy, _dstats=None):
        x, y, gamma, stats = ctx.saved_tensors
        N, C, H, W = x.shape
        M = N * H * W
        dev = x.device
        relu_code = ctx.relu
        tok = ctx.out_token
        if tok is not None and tok.masked_grad is not None:
            # the norm-add-ReLU that took this norm's output as its residual handed its incoming gradient over
            # unmasked, with the mask bits: it must arrive here as that very tensor, untouched
            handed, version, mask = tok.masked_grad
            tok.masked_grad = None
            done, tok.done = tok.done, None
            if (dy.data_ptr() != handed.data_ptr() or dy._version != version or dy.shape != handed.shape):
                raise RuntimeError("batch_norm_act: the output of a plain batch norm fed a norm-add-ReLU residual AND "
                                   "another consumer; the masked gradient hand-over needs the residual to be its only "
                                   "use (ops.conv.set_bn_stats(False) disables the hand-over)")
            if done is not None:
                # the norm-add-ReLU already ran this norm's backward together with its own
                if ctx.request is not None:
                    r = ctx.request
                    r.x = r.stats = r.mask = r.partials = r.dy_ref = None
                return (done[0], done[1], done[2]) + (None,) * 12
            if dy.is_contiguous(memory_format=torch.channels_last) and dy.dtype == torch.bfloat16 and dy.shape == x.shape:
                relu_code, y = 3, mask
            else:
                from .conv import _apply_mask_bits
                dy = _apply_mask_bits(dy, mask)
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        need_dres = ctx.has_res and ctx.relu != 0
        token = ctx.fork_token
        if need_dres and token is not None and dy.dtype == torch.bfloat16:
            need_dres = False             # the residual branch receives dy and the mask (see ForkToken)
        else:
            token = None
        dres = torch.empty_like(x) if need_dres else None
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        coef = torch.empty(3, C, dtype=torch.float32, device=dev)
        lib = _lib.kernels()
        nb = lib.dtf_bn_num_partials(ctypes.c_longlong(M), C)
        ws = _scratch(dev, nb * 2 * C)
        req = ctx.request
        owner = token.owner if token is not None else None
        if (_dual_bwd and owner is not None and req is not None and req.matches(dy) and relu_code == 3
                and owner[0].shape == x.shape and owner[0].dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
                and owner[0].is_contiguous(memory_format=torch.channels_last)):
            # closing norm of a down-sampling block: the branch norm's backward runs here too (see ForkToken)
            z, gamma2, stats2 = owner
            part = req.partials
            fold = max(nb, 512) * 2 * C
            ws = _scratch(dev, fold + nb * 2 * C)
            dz = torch.empty_like(z)
            dgb2 = torch.empty(2, C, dtype=torch.float32, device=dev)
            coef2 = torch.empty(3, C, dtype=torch.float32, device=dev)
            rc = lib.dtf_bn_bwd_stats_dual(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(stats[0]),
                                           _lib.ptr(stats[1]), _lib.ptr(dx), _lib.ptr(dgb[0]), _lib.ptr(dgb[1]),
                                           _lib.ptr(coef), _lib.ptr(part), int(part.shape[0]), _lib.ptr(ws),
                                           _lib.ptr(z), _lib.ptr(gamma2), _lib.ptr(stats2[0]), _lib.ptr(stats2[1]),
                                           _lib.ptr(dz), _lib.ptr(dgb2[0]), _lib.ptr(dgb2[1]), _lib.ptr(coef2),
                                           ctypes.c_void_p(ws.data_ptr() + 4 * fold), ctypes.c_longlong(M), C,
                                           _reduce_counters(dev, C), _lib.stream_ptr(dev))
            token.done = (dz, dgb2[0], dgb2[1])
            global dual_launches
            dual_launches += 1
        elif req is not None and req.matches(dy):
            # the convolution that consumed the output already reduced this very gradient
            part = req.partials
            ws = _scratch(dev, max(nb, 512) * 2 * C)
            rc = lib.dtf_bn_bwd_stats(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(stats[0]),
                                      _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), _lib.ptr(dx),
                                      _lib.ptr(dres), _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(coef),
                                      _lib.ptr(part), int(part.shape[0]), _lib.ptr(ws), ctypes.c_longlong(M), C,
                                      int(relu_code), _reduce_counters(dev, C), _lib.stream_ptr(dev))
        else:
            rc = lib.dtf_bn_bwd(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(stats[0]),
                                _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), _lib.ptr(dx),
                                _lib.ptr(dres), _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(coef), _lib.ptr(ws),
                                ctypes.c_longlong(M), C, int(relu_code), _lib.stream_ptr(dev))
        if req is not None:
            req.x = req.stats = req.mask = req.partials = req.dy_ref = None
        _lib.check(rc, "bn_bwd")
        if ctx.has_res and ctx.relu == 0:
            dres = dy
        if token is not None:
            token.masked_grad = (dy, dy._version, y)     # y holds the mask bytes saved by the forward pass
            dres = dy
        return dx, dgb[0], dgb[1], None, None, None, None, None, dres, None, None, None, None, None, None


def batch_norm_deferred(x, gamma, beta, running_mean, running_var, momentum: float = 0.1, eps: float = 1e-5,
                        relu: bool = True, partials=None):
    """Training-mode ``relu(batch_norm(x))`` WITHOUT its apply pass: the statistics are finished from ``partials``
    (the sums the producing convolution gathered) and a :class:`DeferredNorm` is returned, or ``None`` when the fast
    path does not apply (the caller then uses :func:`batch_norm_act`)."""
    if not (_deferred and partials is not None and _fast_path(x) and gamma.dtype == torch.float32
            and torch.is_grad_enabled()):
        return None
    req = GradStatsRequest() if x.requires_grad else None
    alias, stats = _BNActFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, relu, None, partials, req,
                                  True)
    if req is not None:
        setattr(alias, _REQ_ATTR, req)
    return DeferredNorm(alias, stats[2], stats[3], relu)


def batch_norm_lazy(x, gamma, beta, running_mean, running_var, momentum: float = 0.1, eps: float = 1e-5,
                    partials=None):
    """Training-mode plain ``batch_norm(x)`` (no ReLU, no residual: the norm of a down-sampling branch) whose only
    use is the residual input of a norm-add-ReLU: the statistics are finished from ``partials`` and a
    :class:`DeferredNorm` is returned that :func:`batch_norm_act` takes as ``residual=``; its apply kernel then reads
    this norm's INPUT and normalises it on the fly, so the branch's normalised tensor is never written or read back.
    Backward is that of the ordinary plain norm (the alias is the autograd handle of its output, and carries the
    :class:`ForkToken` through which the norm-add-ReLU hands over its incoming gradient with the mask bits).
    ``None`` when the fast path does not apply (the caller then uses :func:`batch_norm_act`)."""
    if not (_lazy_residual and partials is not None and _fast_path(x) and gamma.dtype == torch.float32
            and torch.is_grad_enabled() and x.requires_grad):
        return None
    from . import conv as _conv
    tok = ForkToken() if _conv._bn_grad_stats else None
    alias, stats = _BNActFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, False, None, partials, None,
                                  True, tok)
    if tok is not None:
        setattr(alias, _TOKEN_ATTR, tok)
    return DeferredNorm(alias, stats[2], stats[3], False)


# DTF_BN_DUAL_BWD=0 runs the backward of a down-sampling branch's norm as a pass of its own (A/B switch)
_dual_bwd = __import__("os").environ.get("DTF_BN_DUAL_BWD", "1") != "0"
dual_launches = 0      # how often the one-pass form ran (tests)
# DTF_BN_LAZY_RESIDUAL=0 writes the down-sampling branch's normalised tensor as before (A/B switch)
_lazy_residual = __import__("os").environ.get("DTF_BN_LAZY_RESIDUAL", "1") != "0"


# Off by default: measured on MI355X (ResNet-50, batch 1024) the step is 90.9 ms with the deferred apply and 88.1 ms
# without. A GEMM stages every operand element once per filter tap and once per output-column tile, so the
# normalisation is recomputed 9-36 times per element in kernels whose vector pipes are busy with address
# arithmetic, while the apply pass it replaces runs once per element at the device's copy rate.
_deferred = __import__("os").environ.get("DTF_BN_DEFER", "0") == "1"


def set_deferred(on: bool) -> None:
    """Switches the deferred norm apply on or off (``DTF_BN_DEFER=1`` in the environment turns it on)."""
    global _deferred
    _deferred = bool(on)


def batch_norm_act(x, gamma, beta, running_mean, running_var, training: bool, momentum: float = 0.1,
                   eps: float = 1e-5, relu: bool = True, residual=None, partials=None):
    """``partials`` (training, fast path only): fp32 ``[rows, 2, C]`` partial sums ``sum(x)``, ``sum(x^2)`` over
    disjoint row sets covering ``x``, as produced by ``ops.conv.conv2d(..., bn_stats=True)``."""
    lazy = None
    if isinstance(residual, DeferredNorm):
        if (training and relu and partials is not None and not residual.relu and _fast_path(x)
                and gamma.dtype == torch.float32 and residual.tensor.shape == x.shape):
            lazy, residual = residual, residual.tensor
        else:
            residual = residual.materialize()
    if _fast_path(x) and gamma.dtype == torch.float32:
        if training:
            # the request rides on the output tensor: ops.conv picks it up from its input (see GradStatsRequest)
            req = GradStatsRequest() if (torch.is_grad_enabled() and x.requires_grad) else None
            tok = None
            if req is not None and not relu and residual is None:
                from . import conv as _conv
                tok = ForkToken() if _conv._bn_grad_stats else None
            if lazy is not None:
                y = _BNActFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual, partials,
                                   req, False, tok, lazy.scale, lazy.shift)
            else:
                y = _BNActFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual, partials,
                                   req, False, tok)
            if req is not None:
                setattr(y, _REQ_ATTR, req)
            if tok is not None:
                setattr(y, _TOKEN_ATTR, tok)
            return y
        scale = gamma * torch.rsqrt(running_var + eps)
        shift = beta - running_mean * scale
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
            residual = residual.contiguous(memory_format=torch.channels_last)
        rc = _lib.kernels().dtf_bn_fwd_apply(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(y), _lib.ptr(scale),
                                             _lib.ptr(shift), ctypes.c_longlong(N * H * W), C, int(relu),
                                             _lib.stream_ptr(x.device))
        _lib.check(rc, "bn_fwd_apply")
        return y
    from . import generic_nn as _gen
    if _gen.usable(x):
        # fp32 activations (the launcher's default dtype) and channel counts the 16-byte kernels do not take: the
        # framework's general kernels, not the vendor library's
        return _gen.batch_norm(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, residual)
    y = F.batch_norm(x.float() if x.dtype != torch.float32 and not x.is_cuda else x, running_mean, running_var,
                     gamma, beta, training, momentum, eps)      # CPU (plumbing mode)
    y = y.to(x.dtype)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class _BNReluPoolFn(torch.autograd.Function):
    """``max_pool(relu(bn(x)))`` without materialising the normalised activation (``csrc/kernels/norm.hip``)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, momentum, eps, k, s, p, partials=None):
        N, C, H, W = x.shape
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dev = x.device
        y = torch.empty((N, OH, OW, C), dtype=x.dtype, device=dev)
        idx = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=dev)
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)
        lib = _lib.kernels()
        nb = lib.dtf_bn_num_partials(ctypes.c_longlong(N * H * W), C)
        ws = _scratch(dev, max(nb, 512) * 2 * C)
        if partials is not None and (partials.dtype != torch.float32 or partials.dim() != 3
                                     or tuple(partials.shape[1:]) != (2, C) or not partials.is_contiguous()):
            raise ValueError("batch_norm_relu_max_pool: partials must be a contiguous fp32 [rows, 2, C] tensor")
        rc = lib.dtf_bn_relu_maxpool_fwd_stats(_lib.ptr(x), _lib.ptr(y), _lib.ptr(idx), _lib.ptr(gamma), _lib.ptr(beta),
                                               _lib.ptr(run_mean), _lib.ptr(run_var), _lib.ptr(stats[0]),
                                               _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]),
                                               _lib.ptr(partials), int(partials.shape[0]) if partials is not None else 0,
                                               _lib.ptr(ws), N, H, W, C, OH, OW, k, s, p, ctypes.c_float(eps),
                                               ctypes.c_float(momentum), _lib.stream_ptr(dev))
        _lib.check(rc, "bn_relu_maxpool_fwd")
        ctx.save_for_backward(x, idx, gamma, stats)
        ctx.geom = (N, C, H, W, OH, OW, k, s, p)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        x, idx, gamma, stats = ctx.saved_tensors
        N, C, H, W, OH, OW, k, s, p = ctx.geom
        dev = x.device
        if not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        coef = torch.empty(3, C, dtype=torch.float32, device=dev)
        lib = _lib.kernels()
        nb = lib.dtf_bn_num_partials(ctypes.c_longlong(N * H * W), C)
        ws = _scratch(dev, nb * 2 * C)
        rc = lib.dtf_bn_relu_maxpool_bwd(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(idx), _lib.ptr(gamma), _lib.ptr(stats[0]),
                                         _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), _lib.ptr(dx),
                                         _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(coef), _lib.ptr(ws), N, H, W, C, OH,
                                         OW, k, s, p, _lib.stream_ptr(dev))
        _lib.check(rc, "bn_relu_maxpool_bwd")
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, None, None


def batch_norm_relu_max_pool(x, gamma, beta, running_mean, running_var, training: bool, momentum: float = 0.1,
                             eps: float = 1e-5, kernel_size: int = 3, stride: int = 2, padding: int = 1, partials=None):
    """``max_pool2d(relu(batch_norm(x)))`` -- the end of the ResNet stem. In training on the bf16 GPU path the
    normalised activation (the widest tensor of the network) is never written: the pool normalises its taps on the
    fly and the norm's backward passes gather the pool's gradient (``DTF_BN_POOL_FUSED=0`` switches this off).
    ``partials``: the statistics of ``x`` as its convolution reduced them (``ops.conv.conv2d(..., bn_stats=True)``);
    the separate reduction pass over ``x`` then does not run."""
    C = x.shape[1]
    if (training and _pool_fused and _fast_path(x) and gamma.dtype == torch.float32 and C <= 2048 and 256 % (C // 8) == 0
            and padding < kernel_size):
        return _BNReluPoolFn.apply(x, gamma, beta, running_mean, running_var, momentum, eps, kernel_size, stride,
                                   padding, partials)
    from . import pool as _pool
    y = batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum, eps, True, None,
                       partials if training else None)
    return _pool.max_pool2d(y, kernel_size, stride, padding)
