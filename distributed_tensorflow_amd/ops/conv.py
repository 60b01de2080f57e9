"""2-D convolution front end for channels_last (NHWC) activations.

Every bf16 convolution on a GPU runs on the hand-written MFMA kernel (``csrc/kernels/gemm.hip``): forward,
data gradient and split-K weight gradient are implicit GEMMs on the NHWC bytes.

* A 1x1 stride-1 convolution (two thirds of ResNet-50's layers) is literally
  ``y[N*H*W, Cout] = x[N*H*W, Cin] @ w^T``; larger kernels and strides gather the shifted / strided pixels
  while the operand is staged. Layers at most 64 channels wide run on the 128 x 64 output tile.
* The data gradient of a 1x1 stride-2 convolution is one plain product over the pixels the forward pass read,
  stored at every other pixel with zeros written beside it (``up2`` of the kernel).
* A first layer (input channels not a multiple of 8: grey-scale or RGB images) is zero-padded on the channel
  axis by a small kernel so that operand chunks are 16 bytes: to 4 channels for a stride-2 layer on an even
  width -- two neighbouring pixels then form one 8-"channel" pixel of a half-width image, and a ``KW``-tap row
  becomes ``ceil((KW + P % 2) / 2)`` taps of stride 1 -- and to 8 channels otherwise.

There is no route to a vendor convolution library: bf16 geometries the implicit-GEMM kernel does not take and fp32
tensors run as patch matrix times weight matrix on the general-purpose exact-fp32 MFMA GEMM (:class:`_ConvAnyFn`);
only CPU tensors (the plumbing mode) use ``torch.nn.functional.conv2d``. ``tools/bench_conv.py`` times the vendor
library next to these kernels by calling it directly.
"""
from __future__ import annotations

import os

import ctypes

import torch
import torch.nn.functional as F

from . import _lib
from . import gemm as _gemm
from . import gemm_any as _any

class _SideGrads:
    """Weight gradients on a second stream.

    A weight gradient is needed by nobody until the gradient bucket it belongs to is packed, while the data gradient
    next to it is on the critical path of the backward pass. Between ``open()`` and ``close()`` (the gradient
    reducer brackets a training step with them) ``run(fn, inputs)`` launches ``fn`` on a side stream that first
    waits for everything the main stream has queued so far, and remembers an event per result; ``wait(tensor)``
    makes the current stream wait for the kernels that produce ``tensor``. The weight-gradient GEMMs (MFMA and issue
    bound) then share the GPU with the memory-bound norm and data-gradient kernels of the layers before them.
    Outside the bracket, during hipGraph capture and on the CPU ``run`` just calls ``fn``.
    """

    def __init__(self):
        self.enabled = os.environ.get("DTF_WGRAD_STREAM", "1") == "1"     # A/B switch; 76.3 -> 74.9 ms per ResNet-50 step
        self._open = False
        self._streams: dict = {}
        self._events: dict = {}
        self._held: list = []

    def open(self) -> None:
        self._open = self.enabled
        self._events.clear()

    def run(self, fn, out: torch.Tensor, keep) -> None:
        """``fn()`` launches the kernels that fill ``out`` (allocated by the caller, on the main stream, so that the
        caching allocator orders its reuse after the main-stream kernels that read it). ``keep``: every tensor the
        kernels read; held here until the side stream is done with them -- a reference, not ``record_stream``,
        which made the allocator of a 100 GB step poll thousands of events and measured five times slower."""
        dev = out.device
        if not (self._open and dev.type == "cuda") or torch.cuda.is_current_stream_capturing():
            fn()
            return
        side = self._streams.get(dev)
        if side is None:
            side = self._streams[dev] = torch.cuda.Stream(dev)      # (a lower stream priority measured no different)
        main = torch.cuda.current_stream(dev)
        side.wait_event(main.record_event())
        with torch.cuda.stream(side):
            fn()
            done = side.record_event()
        self._events[out.data_ptr()] = done
        self._held = [h for h in self._held if not h[0].query()]
        self._held.append((done, keep, out))

    def wait(self, t: torch.Tensor) -> None:
        ev = self._events.pop(t.data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream(t.device).wait_event(ev)

    def close(self) -> None:
        """Every outstanding result becomes visible to the current streams."""
        for dev, side in self._streams.items():
            if self._events or self._held:
                torch.cuda.current_stream(dev).wait_stream(side)
        self._events.clear()
        # the main stream now runs after everything the side stream did: frees are ordered behind it
        self._held.clear()
        self._open = False


side_grads = _SideGrads()
_SIDE_MIN_ELEMS = int(os.environ.get("DTF_WGRAD_SIDE_MIN", str(1 << 25)))      # operand elements (gradient + activation) from which a weight gradient takes the side stream


class _ConvFn(torch.autograd.Function):
    """k x k convolution (stride 1 or 2) as three implicit-GEMM launches on the NHWC bytes: the image
    operand is gathered pixel by pixel while it is staged into LDS, so there is no im2col buffer, no
    gathered copy for strides and no layout change. ``grad_add`` (optional) is a tensor that the data
    gradient is accumulated onto inside the GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, auto=False, stats=None, grad_req=None, bias=None, activation=None,
                out_hw=None, in_scale=None, in_shift=None, in_relu=False):
        """``in_scale`` / ``in_shift`` / ``in_relu``: ``x`` is the INPUT of a batch norm whose apply pass was deferred
        (``ops.norm.DeferredNorm``); the product normalises (and rectifies) it while staging, here and in the weight
        gradient. ``bias`` / ``activation`` ('relu'): applied in the GEMM epilogue. ``out_hw``: output size when it is not
        ``(H + 2 * padding - KH) // stride + 1`` -- ``padding`` counts at the top / left, the rest of an asymmetric
        ('same') padding follows from the output size.
        ``grad_req``: the ``ops.norm.GradStatsRequest`` of the batch norm that produced ``x``, if any.
        ``stats`` (optional, filled in place): fp32 ``[gemm.col_stats_rows(M, K), 2, K]`` partial batch-norm
        statistics of the output, accumulated in the GEMM's store phase."""
        N, C, H, W = x.shape
        K, _, KH, KW = w.shape
        OH, OW = out_hw if out_hw is not None else ((H + 2 * padding - KH) // stride + 1,
                                                    (W + 2 * padding - KW) // stride + 1)
        y = torch.empty((N, OH, OW, K), dtype=x.dtype, device=x.device)
        M = N * OH * OW
        ctx.in_bn = in_bn = (in_scale, in_shift, in_relu) if in_scale is not None else None
        if KH == 1 and KW == 1 and stride == 1 and padding == 0:
            _gemm.gemm_raw(_gemm.NT, x, w, y, M, K, C, C, C, K, col_stats=stats, bias=bias, activation=activation,
                           in_bn=in_bn)
        else:
            ex = _gemm.conv_geom(_gemm.CONV_FWD, H, W, OH, OW, stride, padding, KH, KW, C, K, N)
            _gemm.gemm_raw(_gemm.NT, x, w, y, M, K, KH * KW * C, C, KH * KW * C, K, extra=ex, col_stats=stats,
                           bias=bias, activation=activation, in_bn=in_bn)
        ctx.save_for_backward(x, w, y if activation == "relu" else None)
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.geom = (N, C, H, W, K, KH, KW, OH, OW, stride, padding)
        ctx.auto = auto
        ctx.grad_req = grad_req
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy, g_alias=None):
        """``g_alias`` (fork variant only): gradient that arrived through the aliased input; it is
        added to the data gradient inside the GEMM epilogue instead of by a separate add kernel."""
        x, w, y_act = ctx.saved_tensors
        N, C, H, W, K, KH, KW, OH, OW, stride, padding = ctx.geom
        if not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.contiguous(memory_format=torch.channels_last)
        gb = None
        if y_act is not None:          # ReLU from the epilogue: mask the incoming gradient by the saved output
            gy = _gemm.relu_bwd(gy.permute(0, 2, 3, 1), y_act).permute(0, 3, 1, 2)
        if ctx.bias_dtype is not None and ctx.needs_input_grad[7]:
            gb = _gemm.col_sum(gy.permute(0, 2, 3, 1).reshape(-1, K), dtype=ctx.bias_dtype)
        # a norm-add-ReLU that took the alias as its residual may have handed over its incoming gradient
        # unmasked, with the mask to apply (ops.norm.ForkToken)
        alias_mask = None
        token = getattr(ctx, "fork_token", None)
        if token is not None and token.masked_grad is not None:
            handed, version, alias_mask = token.masked_grad
            token.masked_grad = None
            # the gradient that arrives through the alias must be the very tensor the norm handed over, untouched:
            # a second consumer of the alias makes autograd sum into a new tensor or in place (version counter)
            if (g_alias is None or g_alias.data_ptr() != handed.data_ptr() or g_alias._version != version
                    or g_alias.shape != handed.shape):
                raise RuntimeError("conv2d_fork: the alias fed a norm-add-ReLU residual AND another consumer; the "
                                   "masked gradient hand-over needs the residual to be its only use "
                                   "(ops.conv.set_bn_stats(False) disables the hand-over)")
            if not (g_alias.dtype == x.dtype and g_alias.shape == x.shape
                    and g_alias.is_contiguous(memory_format=torch.channels_last)
                    and ctx.needs_input_grad[0] and not up2_geom(KH, KW, stride, padding, H, W)):
                g_alias, alias_mask = _apply_mask_bits(g_alias, alias_mask), None
        plain = KH == 1 and KW == 1 and stride == 1 and padding == 0
        # 1x1, stride 2: the pixels that were read receive dy @ w, their three neighbours zeros
        up2 = (KH == 1 and KW == 1 and stride == 2 and padding == 0 and H % 2 == 0 and W % 2 == 0
               and N * OH * OW < (1 << 24))
        gx = gw = None
        want_x, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if want_x and up2:
            gx = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
            ex = _gemm.GemmExtra()
            ex.nb1 = ex.nb2 = 1
            ex.up2, ex.H, ex.W, ex.NB = 1, H, W, N
            _gemm.gemm_raw(_gemm.NN, gy, w, gx, N * OH * OW, C, K, K, C, C, extra=ex)
            gx = gx.permute(0, 3, 1, 2)
            want_x = False
        if want_x:
            gx = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
            add = add_mask = None
            if (g_alias is not None and g_alias.dtype == gx.dtype and g_alias.shape == x.shape
                    and g_alias.is_contiguous(memory_format=torch.channels_last)):
                add, g_alias, add_mask = g_alias.permute(0, 2, 3, 1), None, alias_mask
            # x came out of a batch norm: reduce that norm's backward sums over gx while gx is stored
            req, part, bn_grad = ctx.grad_req, None, None
            if (_bn_grad_stats and req is not None and req.consumers == 1 and req.x is not None and stride == 1
                    and req.x.shape == x.shape
                    and req.x.dtype == torch.bfloat16 and req.x.is_contiguous(memory_format=torch.channels_last)):
                part = torch.empty((_gemm.col_stats_rows(N * H * W, C), 2, C), dtype=torch.float32, device=x.device)
                bn_grad = (req.x, req.stats, req.mask, req.relu)
            if plain:
                _gemm.gemm_raw(_gemm.NN, gy, w, gx, N * H * W, C, K, K, C, C, add=add, col_stats=part, bn_grad=bn_grad,
                               add_mask=add_mask)
            else:
                ex = _gemm.conv_geom(_gemm.CONV_DGRAD, H, W, OH, OW, stride, padding, KH, KW, C, K, N)
                _gemm.gemm_raw(_gemm.NN, gy, w, gx, N * H * W, C, KH * KW * K, K, KH * KW * C, C, extra=ex, add=add,
                               col_stats=part, bn_grad=bn_grad, add_mask=add_mask)
            gx = gx.permute(0, 3, 1, 2)
            if part is not None:
                req.hand_over(part, gx)
        if g_alias is not None and ctx.needs_input_grad[0]:
            gx = g_alias if gx is None else gx + g_alias
        if want_w:
            in_bn = ctx.in_bn         # the activation operand is a deferred norm's input: normalised while staged

            transposed = (not plain) and K <= 64 and _wgrad_t
            gw = torch.empty((K, KH, KW, C), dtype=w.dtype, device=x.device)
            gwt = torch.empty((KH * KW * C, K), dtype=w.dtype, device=x.device) if transposed else None

            def weight_gradient():
                if plain:
                    _gemm.gemm_raw(_gemm.TN, gy, x, gw, K, C, N * OH * OW, K, C, C, in_bn=in_bn)
                elif transposed:
                    # narrow layer: the transposed product, every tap side by side in the rows (dy is re-read once
                    # per 128 of them instead of once per tap)
                    ex = _gemm.conv_geom(_gemm.CONV_WGRAD_T, H, W, OH, OW, stride, padding, KH, KW, C, K, N)
                    _gemm.gemm_raw(_gemm.TN, x, gy, gwt, KH * KW * C, K, N * OH * OW, C, K, K, extra=ex, in_bn=in_bn)
                    gw.view(K, KH * KW * C).copy_(gwt.t())
                else:
                    ex = _gemm.conv_geom(_gemm.CONV_WGRAD, H, W, OH, OW, stride, padding, KH, KW, C, K, N)
                    _gemm.gemm_raw(_gemm.TN, gy, x, gw, K, C, N * OH * OW, K, C, KH * KW * C, extra=ex, in_bn=in_bn)

            # (on the side stream only when autograd will keep the very tensor as the parameter's gradient: a weight
            # in another memory format gets a converted copy, made on the main stream before the side stream has
            # written the source)
            # ... and only for products large enough to be worth two event calls on the host: a small model's step is
            # host bound (ResNet-20 at batch 128: 3.7 ms per step without, 5.1 ms with the side stream)
            if w.is_contiguous(memory_format=torch.channels_last) and gy.numel() + x.numel() >= _SIDE_MIN_ELEMS:
                side_grads.run(weight_gradient, gw, (gy, x, gwt, in_bn))
            else:
                weight_gradient()
            gw = gw.permute(0, 3, 1, 2)        # [K, C, KH, KW] with channels_last strides
        return gx, gw, None, None, None, None, None, gb, None, None, None, None, None


def up2_geom(KH, KW, stride, padding, H, W) -> bool:
    return KH == 1 and KW == 1 and stride == 2 and padding == 0 and H % 2 == 0 and W % 2 == 0


class _ConvForkFn(_ConvFn):
    """``(conv(x, w), alias of x)``: the block input feeds both the first convolution and the identity
    (or down-sampling) branch. Routing the second use through the alias hands its gradient to this
    function's backward, which accumulates it in the epilogue of the data-gradient GEMM; autograd's
    separate add kernel over the largest activations of the network disappears."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, auto=False, stats=None, grad_req=None, token=None):
        y = _ConvFn.forward(ctx, x, w, stride, padding, auto, stats, grad_req)
        ctx.fork_token = token
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, g_alias):
        if gy is None:       # the convolution branch was unused
            tk = ctx.fork_token
            if tk is not None and tk.masked_grad is not None:
                g_alias, tk.masked_grad = _apply_mask_bits(g_alias, tk.masked_grad[2]), None
            return g_alias, None, None, None, None, None, None, None
        return _ConvFn.backward(ctx, gy, g_alias)[:7] + (None,)


def pad_channels(x: torch.Tensor, cp: int) -> torch.Tensor:
    """``[N, H, W, cp]`` copy of a channels_last bf16 ``[N, C, H, W]`` activation with zeros in channels ``C..cp``."""
    N, C, H, W = x.shape
    if C == cp:
        return x.permute(0, 2, 3, 1)
    out = torch.empty((N, H, W, cp), dtype=x.dtype, device=x.device)
    rc = _lib.kernels().dtf_pad_channels_bf16(_lib.ptr(x), _lib.ptr(out), ctypes.c_longlong(N * H * W), C, cp,
                                              _lib.stream_ptr(x.device))
    _lib.check(rc, "pad_channels_bf16")
    return out


class _FirstConvFn(torch.autograd.Function):
    """Convolution of an image with few channels (not a multiple of 8), see the module docstring: the input is
    channel-padded once, the weights are padded (and, in the pixel-pair form, shifted) to match, and forward and
    weight gradient run on the implicit-GEMM kernel; the batch is cut into chunks that respect the kernel's index
    ranges (2^24 GEMM rows, 2^31 operand elements)."""

    @staticmethod
    def _geometry(C, H, W, KH, KW, stride, padding):
        pair = C <= 4 and stride == 2 and W % 2 == 0
        if pair:
            shift = padding & 1
            return dict(pair=True, cp=4, shift=shift, kwv=(KW + shift + 1) // 2, civ=8, wv=W // 2, swv=1,
                        pwv=(padding + shift) // 2)
        cp = -(-C // 8) * 8
        return dict(pair=False, cp=cp, shift=0, kwv=KW, civ=cp, wv=W, swv=stride, pwv=padding)

    @staticmethod
    def _pad_weight(w, g):
        K, C, KH, KW = w.shape
        taps = 2 * g["kwv"] if g["pair"] else KW
        wp = torch.zeros((K, KH, taps, g["cp"]), dtype=w.dtype, device=w.device)
        wp[:, :, g["shift"]:g["shift"] + KW, :C] = w.permute(0, 2, 3, 1)
        return wp

    @staticmethod
    def forward(ctx, x, w, stride, padding, stats=None, bias=None, activation=None, out_hw=None):
        N, C, H, W = x.shape
        K, _, KH, KW = w.shape
        OH, OW = out_hw if out_hw is not None else ((H + 2 * padding - KH) // stride + 1,
                                                    (W + 2 * padding - KW) // stride + 1)
        g = _FirstConvFn._geometry(C, H, W, KH, KW, stride, padding)
        xp = pad_channels(x, g["cp"])
        wp = _FirstConvFn._pad_weight(w, g)
        kdim = KH * g["kwv"] * g["civ"]
        y = torch.empty((N, OH, OW, K), dtype=x.dtype, device=x.device)
        nb = max(1, min(((1 << 24) - 1) // max(OH * OW, 1), ((1 << 31) - 1) // (H * W * g["cp"]),
                        ((1 << 31) - 1) // (OH * OW * K)))
        rows = 0
        for b0 in range(0, N, nb):
            b1 = min(N, b0 + nb)
            M = (b1 - b0) * OH * OW
            ex = _gemm.conv_geom(_gemm.CONV_FWD, H, g["wv"], OH, OW, stride, padding, KH, g["kwv"], g["civ"], K,
                                 b1 - b0, stride_w=g["swv"], pad_w=g["pwv"])
            st = None
            if stats is not None:
                r = _gemm.col_stats_rows(M, K)
                st = stats[rows:rows + r]
                rows += r
            _gemm.gemm_raw(_gemm.NT, xp[b0:b1], wp, y[b0:b1], M, K, kdim, g["civ"], kdim, K, extra=ex, col_stats=st,
                           bias=bias, activation=activation)
        ctx.save_for_backward(xp, w, y if activation == "relu" else None)
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.geom = (N, C, H, W, K, KH, KW, OH, OW, stride, padding, nb)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def stats_rows(x, w, stride, padding) -> int:
        """Rows of the statistics buffer ``forward`` fills (the batch chunks' partial rows, one after the other)."""
        N, C, H, W = x.shape
        K, _, KH, KW = w.shape
        OH = (H + 2 * padding - KH) // stride + 1
        OW = (W + 2 * padding - KW) // stride + 1
        cp = 4 if (C <= 4 and stride == 2 and W % 2 == 0) else -(-C // 8) * 8
        nb = max(1, min(((1 << 24) - 1) // max(OH * OW, 1), ((1 << 31) - 1) // (H * W * cp),
                        ((1 << 31) - 1) // (OH * OW * K)))
        return sum(_gemm.col_stats_rows((min(N, b0 + nb) - b0) * OH * OW, K) for b0 in range(0, N, nb))

    @staticmethod
    def backward(ctx, gy):
        xp, w, y_act = ctx.saved_tensors
        N, C, H, W, K, KH, KW, OH, OW, stride, padding, nb = ctx.geom
        g = _FirstConvFn._geometry(C, H, W, KH, KW, stride, padding)
        if not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.contiguous(memory_format=torch.channels_last)
        gyn = gy.permute(0, 2, 3, 1)
        if y_act is not None:
            gyn = _gemm.relu_bwd(gyn, y_act)
        gx = gw = gb = None
        if ctx.bias_dtype is not None and ctx.needs_input_grad[5]:
            gb = _gemm.col_sum(gyn.reshape(-1, K), dtype=ctx.bias_dtype)
        if ctx.needs_input_grad[1]:
            ncol = g["kwv"] * g["civ"]
            fold = ncol <= 64
            chunks = range(0, N, nb)
            acc = None
            for b0 in chunks:
                b1 = min(N, b0 + nb)
                pdt = w.dtype if len(chunks) == 1 else torch.float32
                if K <= 64 and _wgrad_t:      # transposed product, every tap side by side in the rows
                    ex = _gemm.conv_geom(_gemm.CONV_WGRAD_T, H, g["wv"], OH, OW, stride, padding, KH, g["kwv"],
                                         g["civ"], K, b1 - b0, stride_w=g["swv"], pad_w=g["pwv"])
                    part = torch.empty((KH * ncol, K), dtype=pdt, device=w.device)
                    _gemm.gemm_raw(_gemm.TN, xp[b0:b1], gyn[b0:b1], part, KH * ncol, K, (b1 - b0) * OH * OW, g["civ"],
                                   K, K, extra=ex)
                    part = part.t()
                else:
                    ex = _gemm.conv_geom(_gemm.CONV_WGRAD, H, g["wv"], OH, OW, stride, padding, KH, g["kwv"], g["civ"],
                                         K, b1 - b0, stride_w=g["swv"], pad_w=g["pwv"], fold=fold)
                    part = torch.empty((K, KH * ncol), dtype=pdt, device=w.device)
                    _gemm.gemm_raw(_gemm.TN, gyn[b0:b1], xp[b0:b1], part, K, ncol if fold else g["civ"],
                                   (b1 - b0) * OH * OW, K, g["civ"], KH * ncol, extra=ex)
                acc = part if acc is None else acc + part
            acc = acc.contiguous()
            taps = 2 * g["kwv"] if g["pair"] else KW
            gw = acc.view(K, KH, taps, g["cp"])[:, :, g["shift"]:g["shift"] + KW, :C].permute(0, 3, 1, 2).to(w.dtype)
        if ctx.needs_input_grad[0]:
            # gradient of the image itself (rare): data gradient against 8-channel padded weights, then cut
            cp8 = -(-C // 8) * 8
            w8 = torch.zeros((K, KH, KW, cp8), dtype=w.dtype, device=w.device)
            w8[..., :C] = w.permute(0, 2, 3, 1)
            gx8 = torch.empty((N, H, W, cp8), dtype=gy.dtype, device=gy.device)
            nbx = max(1, min(((1 << 24) - 1) // (H * W), ((1 << 31) - 1) // (OH * OW * K)))
            for b0 in range(0, N, nbx):
                b1 = min(N, b0 + nbx)
                ex = _gemm.conv_geom(_gemm.CONV_DGRAD, H, W, OH, OW, stride, padding, KH, KW, cp8, K, b1 - b0)
                _gemm.gemm_raw(_gemm.NN, gyn[b0:b1], w8, gx8[b0:b1], (b1 - b0) * H * W, cp8, KH * KW * K, K,
                               KH * KW * cp8, cp8, extra=ex)
            gx = gx8[..., :C].permute(0, 3, 1, 2)
        return gx, gw, None, None, None, gb, None, None


def _gemm_ok(x: torch.Tensor, w: torch.Tensor, stride: int = 1) -> bool:
    pixels = x.shape[0] * x.shape[2] * x.shape[3]
    # output pixels, bounded from above for any padding below the kernel size
    out_pixels = x.shape[0] * ((x.shape[2] + w.shape[2] - 2) // stride + 1) * ((x.shape[3] + w.shape[3] - 2) // stride + 1)
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.shape[1] % 8 == 0
            and w.shape[0] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and w.is_contiguous(memory_format=torch.channels_last)
            and pixels < (1 << 24)
            # operands are addressed with 32-bit byte offsets (buffer descriptors): below 2^31 elements each
            and pixels * x.shape[1] < (1 << 31) and out_pixels * w.shape[0] < (1 << 31))


_bn_stats = os.environ.get("DTF_CONV_BN_STATS", "1") != "0"
_wgrad_t = os.environ.get("DTF_CONV_WGRAD_T", "1") != "0"     # transposed weight gradient of narrow layers (A/B switch)


_bn_grad_stats = os.environ.get("DTF_CONV_BN_GRAD_STATS", "1") != "0"


def set_bn_stats(on: bool, grad: bool | None = None) -> None:
    """Switches the hand-over of partial batch-norm statistics between convolution and norm (A/B measurements):
    ``on`` the forward sums a convolution gathers for the norm after it, ``grad`` (default: same as ``on``) the
    backward sums its data-gradient GEMM gathers for the norm before it."""
    global _bn_stats, _bn_grad_stats
    _bn_stats = bool(on)
    _bn_grad_stats = bool(on if grad is None else grad)


def _apply_mask_bits(g: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``g`` with the elements whose bit in ``mask`` (one bit per element in memory order) is 0 set to zero --
    the slow path of the masked gradient hand-over, for gradients that cannot ride the GEMM epilogue."""
    gm = g.contiguous(memory_format=torch.channels_last) if g.dim() == 4 else g.contiguous()
    flat = gm.permute(0, 2, 3, 1).reshape(-1, 8) if g.dim() == 4 else gm.reshape(-1, 8)
    bits = (mask.reshape(-1, 1) >> torch.arange(8, device=mask.device, dtype=torch.uint8)) & 1
    out = (flat * bits.to(flat.dtype)).reshape(gm.permute(0, 2, 3, 1).shape if g.dim() == 4 else gm.shape)
    return out.permute(0, 3, 1, 2) if g.dim() == 4 else out


def _norm_request(x):
    from . import norm as _norm
    return _norm.request_of(x) if (_bn_grad_stats and torch.is_grad_enabled()) else None


def _own_path(x, w, stride) -> bool:
    return (stride in (1, 2) and _gemm_ok(x, w, stride)
            and (stride == 1 or (x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)))


def _first_layer_path(x, w, stride) -> bool:
    """Few-channel image input (see :class:`_FirstConvFn`)."""
    return (stride in (1, 2) and x.is_cuda and x.dtype == torch.bfloat16
            and w.dtype == torch.bfloat16 and x.shape[1] % 8 != 0 and x.shape[1] <= 64 and w.shape[0] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last)
            and (stride == 1 or (x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)))


def _stats_buffer(x, w, stride, padding):
    N, _, H, W = x.shape
    K, _, KH, KW = w.shape
    M = N * ((H + 2 * padding - KH) // stride + 1) * ((W + 2 * padding - KW) // stride + 1)
    return torch.empty((_gemm.col_stats_rows(M, K), 2, K), dtype=torch.float32, device=x.device)


def conv2d(x: torch.Tensor, w: torch.Tensor, stride: int = 1, padding: int = 0, bn_stats: bool = False):
    """Bias-free convolution of a channels_last activation (or of an ``ops.norm.DeferredNorm``: the norm is then
    applied while the operand is staged) with an ``[O, I, kh, kw]`` weight.
    ``bn_stats=True`` returns ``(y, partials)``: ``partials`` are the per-channel sums and sums of squares of
    ``y`` over disjoint row sets, accumulated while the GEMM stores ``y`` (pass them to
    ``ops.norm.batch_norm_act(..., partials=)``, which then reads ``y`` once instead of twice), or ``None`` when
    the convolution did not run on the framework's own kernel."""
    from . import norm as _norm
    if isinstance(x, _norm.DeferredNorm):
        dn, x = x, x.tensor
        if _own_path(x, w, stride):
            st = _stats_buffer(x, w, stride, padding) if (bn_stats and _bn_stats) else None
            y = _ConvFn.apply(x, w, stride, padding, True, st, _norm_request(x), None, None, None,
                              dn.scale, dn.shift, dn.relu)
            return (y, st) if bn_stats else y
        x = dn.materialize()
    if _own_path(x, w, stride):
        st = _stats_buffer(x, w, stride, padding) if (bn_stats and _bn_stats) else None
        y = _ConvFn.apply(x, w, stride, padding, True, st, _norm_request(x))
        return (y, st) if bn_stats else y
    if _first_layer_path(x, w, stride):
        st = None
        if bn_stats and _bn_stats:
            st = torch.empty((_FirstConvFn.stats_rows(x, w, stride, padding), 2, w.shape[0]), dtype=torch.float32,
                             device=x.device)
        y = _FirstConvFn.apply(x, w, stride, padding, st)
        return (y, st) if bn_stats else y
    if x.dtype != w.dtype:
        x = x.to(w.dtype)
    if x.is_cuda and _any.usable(x, w):
        # fp32 layers and whatever geometry is left: patch matrix x weight matrix on the general MFMA GEMM
        K, _, KH, KW = w.shape
        OH = (x.shape[2] + 2 * padding - KH) // stride + 1
        OW = (x.shape[3] + 2 * padding - KW) // stride + 1
        wc = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)
        y = _ConvAnyFn.apply(x, wc, None, stride, stride, padding, padding, OH, OW, None)
        return (y, None) if bn_stats else y
    y = F.conv2d(x, w, None, stride, padding)        # CPU (plumbing mode)
    return (y, None) if bn_stats else y


def conv2d_fork(x: torch.Tensor, w: torch.Tensor, stride: int = 1, padding: int = 0, bn_stats: bool = False):
    """``(conv2d(x, w), x')`` where ``x'`` aliases ``x``: use ``x'`` for the other consumer of ``x`` (a
    residual branch) so that the two gradients of ``x`` are summed inside the data-gradient GEMM.
    With ``bn_stats=True`` the result is ``(y, x', partials)`` (see :func:`conv2d`)."""
    if _own_path(x, w, stride) and x.requires_grad and torch.is_grad_enabled():
        st = _stats_buffer(x, w, stride, padding) if (bn_stats and _bn_stats) else None
        from . import norm as _norm
        token = _norm.ForkToken() if (_bn_grad_stats and w.shape[2] == 1 and w.shape[3] == 1 and stride == 1) else None
        y, xi = _ConvForkFn.apply(x, w, stride, padding, True, st, _norm_request(x), token)
        if token is not None:
            setattr(xi, _norm._TOKEN_ATTR, token)
        return (y, xi, st) if bn_stats else (y, xi)
    if bn_stats:
        y, st = conv2d(x, w, stride, padding, True)
        return y, x, st
    return conv2d(x, w, stride, padding), x


class _ConvAnyFn(torch.autograd.Function):
    """Convolution on the general-purpose exact-fp32 MFMA GEMM: the GPU path of fp32 layers and of geometries the bf16
    implicit-GEMM kernel does not take (rectangular strides, odd channel counts deeper in a network). All three passes
    are IMPLICIT products (``ops.gemm_any.conv_product``): the kernel gathers the patch matrix from the NHWC image
    while it stages the operand, so no im2col buffer -- KH * KW times the activation, per pass -- exists."""

    @staticmethod
    def forward(ctx, x, w, bias, sh, sw, pt, pl, OH, OW, activation):
        N, C, H, W = x.shape
        K, _, KH, KW = w.shape
        xn = x.permute(0, 2, 3, 1).contiguous()
        wm = w.permute(0, 2, 3, 1).reshape(K, KH * KW * C)
        act = "relu" if activation == "relu" else None
        y = _any.conv_product("fwd", xn, wm.t(), (H, W, C, OH, OW, KH, KW, sh, sw, pt, pl), N * OH * OW, K,
                              KH * KW * C, bias=bias, activation=act, out_dtype=x.dtype)
        ctx.save_for_backward(xn, w, y if act == "relu" else None)
        ctx.geom = (N, C, H, W, K, KH, KW, OH, OW, sh, sw, pt, pl)
        ctx.bias_dtype = bias.dtype if bias is not None else None
        return y.view(N, OH, OW, K).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        xn, w, y_act = ctx.saved_tensors
        N, C, H, W, K, KH, KW, OH, OW, sh, sw, pt, pl = ctx.geom
        g = gy.permute(0, 2, 3, 1).reshape(-1, K)
        if not g.is_contiguous():
            g = g.contiguous()
        if g.dtype != xn.dtype:
            g = g.to(xn.dtype)
        if y_act is not None:
            g = _any.ew_f32(0, g, y_act) if g.dtype == torch.float32 else _gemm.relu_bwd(g, y_act)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # dx[(n, ih, iw), c] = sum over (kh, kw, ko) of dy at the output pixel that tap reads from, times w
            wd = w.permute(2, 3, 0, 1).reshape(KH * KW * K, C)          # [(kh, kw, ko), c]; the weight is small
            if not wd.is_contiguous():
                wd = wd.contiguous()
            gxn = _any.conv_product("dgrad", g.view(N, OH, OW, K), wd, (OH, OW, K, H, W, KH, KW, sh, sw, pt, pl),
                                    N * H * W, C, KH * KW * K, out_dtype=xn.dtype)
            gx = gxn.view(N, H, W, C).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = _any.conv_product("wgrad", xn, g.t(), (H, W, C, OH, OW, KH, KW, sh, sw, pt, pl), K, KH * KW * C,
                                   N * OH * OW, out_dtype=w.dtype).view(K, KH, KW, C).permute(0, 3, 1, 2)
        if ctx.bias_dtype is not None and ctx.needs_input_grad[2]:
            gb = g.sum(0, dtype=torch.float32).to(ctx.bias_dtype)
        return gx, gw, gb, None, None, None, None, None, None, None


def conv2d_bias_act(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, strides=(1, 1),
                    pads=(0, 0, 0, 0), activation: str | None = None) -> torch.Tensor:
    """Keras-style convolution layer ``act(conv(x, w) + bias)`` on a channels_last activation. ``pads`` = (left,
    right, top, bottom), as TensorFlow's 'same' computes them (the odd pixel at the end); ``activation``: None or
    'relu' (fused), anything else is left to the caller. On a GPU every case runs on the framework's own kernels:
    bf16 with square strides on the implicit-GEMM kernel (bias and ReLU in its epilogue), the rest as patch matrix
    times weight matrix on the general-purpose MFMA GEMM."""
    sh, sw = strides
    pl, pr, pt, pb = pads
    N, C, H, W = x.shape
    K, _, KH, KW = w.shape
    OH = (H + pt + pb - KH) // sh + 1
    OW = (W + pl + pr - KW) // sw + 1
    act = "relu" if activation == "relu" else None
    if x.is_cuda and w.is_cuda:
        if x.dtype != w.dtype:
            x = x.to(w.dtype)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        wc = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)
        square = sh == sw and pt == pl
        if square and x.dtype == torch.bfloat16 and N * max(H * W, OH * OW) < (1 << 24):
            if _own_path(x, wc, sh):
                return _ConvFn.apply(x, wc, sh, pt, False, None, None, bias, act, (OH, OW))
            if _first_layer_path(x, wc, sh):
                return _FirstConvFn.apply(x, wc, sh, pt, None, bias, act, (OH, OW))
        if x.dtype in (torch.float32, torch.bfloat16):
            return _ConvAnyFn.apply(x, wc, bias, sh, sw, pt, pl, OH, OW, act)
    if pl == pr and pt == pb:
        y = F.conv2d(x, w, bias, (sh, sw), (pt, pl))
    else:
        y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, bias, (sh, sw))
    return F.relu(y) if act == "relu" else y
