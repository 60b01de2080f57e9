"""Dense (fully connected) layer op: ``y = act(x @ W^T + b)``.

On a GPU with bf16 operands the forward, the data gradient and the weight
gradient run on the hand-written MFMA GEMM (``ops/gemm.py``) with bias and
ReLU/GELU applied in the epilogue, and the bias gradient is a fused column
reduction; fp32 operands, and bf16 ones whose rows are not 16-byte aligned, run
on the general-purpose exact-fp32 MFMA GEMM (``ops/gemm_any.py``), also with
bias and activation in its epilogue. On CPU the same maths runs in plain torch. This is the MatMul + BiasAdd + ReLU chain of the
reference model (reference initializer.py:16-18).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import gemm as _gemm
from . import gemm_any as _any
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
            gb = _bias_grad(gy, g, ctx.bias_dtype) if ctx.drop is None and ctx.activation is None \
                else _gemm.col_sum(g, dtype=ctx.bias_dtype)
        if gx is not None:
            gx = gx.reshape(ctx.x_shape)
            if g_alias is not None:
                gx = gx + g_alias
        return gx, gw, gb, None, None


def _bias_grad(gy: torch.Tensor, g: torch.Tensor, dtype) -> torch.Tensor:
    """Column sums of the output gradient ``g`` (``gy`` as a matrix): taken from the LayerNorm backward that
    produced ``gy`` when it summed them on the way (``ops.layernorm.take_col_sum``), else a pass over ``g``."""
    if gy.is_cuda and gy.dtype == torch.bfloat16:
        from . import layernorm as _ln
        cs = _ln.take_col_sum(gy)
        if cs is not None:
            return cs if cs.dtype == dtype else cs.to(dtype)
    return _gemm.col_sum(g, dtype=dtype)


def _padded_cols(n: int) -> int:
    return (n + 7) // 8 * 8


# DTF_PADN_BIG=0: wide unaligned layers stay on the 128x128 kernel (A/B switch)
_pad_big = __import__("os").environ.get("DTF_PADN_BIG", "1") != "0"


class _LinearPadNFn(torch.autograd.Function):
    """bf16 layer whose output width is not a multiple of 8 (BERT's 30522-word vocabulary projection): the output
    lives in a buffer with rows padded to 16 bytes and the MFMA kernel runs all three products. Forward: rows of the
    weight past N read as zeros (range check of the buffer descriptor) and whole 16-byte chunks are stored, so the
    padding columns hold zeros. Backward: the gradient arrives in the same padded layout with zeroed padding (the
    fused softmax-cross-entropy produces it so) or is copied into it; the data gradient then contracts over N reading
    whole chunks of the gradient against zero rows of the weight, and the weight gradient stores N rows."""

    @staticmethod
    def _big(M, N, K) -> bool:
        """Large products (the 30522-word projection over tens of thousands of positions) run on the 256x256 kernel
        against a copy of the weight with zero rows up to the padded width (47 MB per step for BERT-base: 20 us,
        against 646 -> ~1,000 TFLOP/s on 4.3 TFLOP of products); the padded width must then also be a whole number
        of 64-deep K-tiles, since the data gradient contracts over it."""
        Np = _padded_cols(N)
        return (_pad_big and Np % 64 == 0 and K % 64 == 0 and _gemm.big_ok(M, Np, K) and _gemm.big_ok(M, K, Np)
                and _gemm._big_splits(Np, K, M) > 0)

    @staticmethod
    def forward(ctx, x, w, b):
        K, N = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, Np = x2.shape[0], _padded_cols(N)
        buf = torch.empty((M, Np), dtype=torch.bfloat16, device=x.device)
        ctx.big = _LinearPadNFn._big(M, N, K) and w.is_contiguous()
        if ctx.big:
            wp = torch.zeros((Np, K), dtype=w.dtype, device=w.device)
            wp[:N].copy_(w)
            bp = None
            if b is not None:
                bp = torch.zeros(Np, dtype=b.dtype, device=b.device)
                bp[:N].copy_(b)
            _gemm.gemm256(_gemm.NT, x2, wp, M, Np, K, bias=bp, out=buf)
            w = wp
        else:
            _gemm.gemm_raw(_gemm.NT, x2, w, buf, M, N, K, K, K, Np, bias=b)
        ctx.bias_dtype = b.dtype if b is not None else None
        ctx.x_shape = x.shape
        ctx.n_out = N
        ctx.save_for_backward(x2, w)
        return buf.view(*x.shape[:-1], Np)[..., :N]

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors          # (big path: w is the zero-padded copy)
        N, K = ctx.n_out, w.shape[1]
        M, Np = x2.shape[0], _padded_cols(N)
        g = gy.reshape(-1, N) if gy.dim() != 2 else gy
        # a [M, N] window of a buffer of exactly M rows of Np elements: the padded layout this function (and the
        # fused loss downstream) produce. Its padding columns are ours to clear.
        owns_padding = (g.dtype == torch.bfloat16 and g.stride() == (Np, 1) and g.storage_offset() == 0
                        and g.untyped_storage().nbytes() == M * Np * 2 and g.data_ptr() % 16 == 0)
        if owns_padding:
            full = g.as_strided((M, Np), (Np, 1))
            full[:, N:].zero_()
        else:
            full = torch.zeros((M, Np), dtype=torch.bfloat16, device=g.device)
            full[:, :N].copy_(g)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if ctx.big:
                gx = _gemm.gemm256(_gemm.NN, full, w, M, K, Np)
            else:
                gx = torch.empty((M, K), dtype=torch.bfloat16, device=g.device)
                _gemm.gemm_raw(_gemm.NN, full, w, gx, M, K, N, Np, K, K)
            gx = gx.reshape(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            if ctx.big:
                gw = _gemm.gemm256(_gemm.TN, full, x2, Np, K, M, splits=_gemm._big_splits(Np, K, M))[:N]
            else:
                gw = torch.empty((N, K), dtype=torch.bfloat16, device=g.device)
                _gemm.gemm_raw(_gemm.TN, full, x2, gw, N, K, M, Np, K, K)
        if ctx.bias_dtype is not None and ctx.needs_input_grad[2]:
            gb = _gemm.col_sum(full, dtype=torch.float32)[:N].to(ctx.bias_dtype)
        return gx, gw, gb


class _LinearAnyFn(torch.autograd.Function):
    """GPU path of everything the bf16 tile kernel does not take (fp32 layers, unaligned rows): the three products
    on the general-purpose MFMA GEMM, which reads the operands through their strides (no transposed copies)."""

    @staticmethod
    def forward(ctx, x, w, b, activation, drop_p=0.0):
        x2 = x.reshape(-1, x.shape[-1])
        odt = torch.float32 if torch.float32 in (x.dtype, w.dtype) else torch.bfloat16
        # ReLU rides the GEMM epilogue; GELU keeps its pre-activation for the backward pass
        y = _any.matmul(x2, w.t(), bias=b, activation="relu" if activation == "relu" else None, out_dtype=odt)
        pre = None
        if activation == "gelu":
            pre = y
            y = _any.ew_f32(2, pre) if odt == torch.float32 else _gemm.gelu_fwd(pre)
        ctx.drop = None
        if drop_p > 0.0:
            st = _rng.state_for(x.device)
            ctx.drop = (drop_p, st.seed, st.snapshot())
            y = _any.ew_f32(3, y, p=drop_p, seed=st.seed, off=ctx.drop[2]) if y.dtype == torch.float32 \
                else _gemm.dropout_fwd(y, *ctx.drop)
            st.advance(y.numel())
        ctx.activation = activation
        ctx.bias_dtype = b.dtype if b is not None else None
        ctx.x_shape = x.shape
        ctx.save_for_backward(x2, w, pre if activation == "gelu" else (y if activation == "relu" else None))
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w, saved = ctx.saved_tensors
        g = gy.reshape(-1, gy.shape[-1])
        if g.dtype == torch.float32:
            if ctx.drop is not None:
                p, seed, off = ctx.drop
                if ctx.activation == "relu":      # y > 0 already implies "kept": only the 1 / (1 - p) scale is needed
                    g = _any.ew_f32(0, g, saved, scale=1.0 / (1.0 - p))
                elif ctx.activation == "gelu":
                    g = _any.ew_f32(1, g, saved, p=p, seed=seed, off=off)
                else:
                    g = _any.ew_f32(3, g, p=p, seed=seed, off=off)
            elif ctx.activation == "relu":
                g = _any.ew_f32(0, g, saved)
            elif ctx.activation == "gelu":
                g = _any.ew_f32(1, g, saved)
        else:
            if ctx.drop is not None:
                g = _gemm.dropout_bwd(g, *ctx.drop, activation=ctx.activation, saved=saved)
            elif ctx.activation == "relu":
                g = _gemm.relu_bwd(g, saved)
            elif ctx.activation == "gelu":
                g = _gemm.gelu_bwd(g, saved)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _any.matmul(g, w, out_dtype=x2.dtype).reshape(ctx.x_shape)            # [M, K] = g [M, N] @ w [N, K]
        if ctx.needs_input_grad[1]:
            gw = _any.matmul(g.t(), x2, out_dtype=w.dtype)                             # [N, K] = g^T [N, M] @ x [M, K]
        if ctx.bias_dtype is not None and ctx.needs_input_grad[2]:
            gb = g.sum(0, dtype=torch.float32).to(ctx.bias_dtype)
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


# DTF_FFN_BIAS_GRAD_FUSED=0: the feed-forward block's first bias gradient from a column-sum pass of its own (A/B switch)
_fused_bias_grad = __import__("os").environ.get("DTF_FFN_BIAS_GRAD_FUSED", "1") != "0"


class _FfnForkFn(torch.autograd.Function):
    """``(gelu(x W1^T + b1) W2^T + b2, alias of x)`` as one node, so that the GELU never runs as a kernel of
    its own: the first GEMM's epilogue writes ``h = gelu(pre)`` and ``d = gelu'(pre)``, and the data-gradient
    GEMM of the second layer multiplies by ``d`` in its epilogue. The alias carries the residual branch's
    gradient into the first layer's data-gradient epilogue (see :class:`_LinearForkFn`)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, H, Fd = x2.shape[0], x2.shape[1], w1.shape[0]
        h = torch.empty((M, Fd), dtype=torch.bfloat16, device=x.device)
        d = torch.empty_like(h)
        _gemm.gemm256(_gemm.NT, x2, w1, M, Fd, H, bias=b1, activation="gelu", out=h, grad_out=d)
        y = _gemm.gemm_nt(h, w2, bias=b2)
        ctx.x_shape = x.shape
        ctx.bias_dtypes = (b1.dtype, b2.dtype)
        ctx.save_for_backward(x2, w1, w2, h, d)
        return y.reshape(*x.shape[:-1], w2.shape[0]), x.view_as(x)

    @staticmethod
    def backward(ctx, gy, g_alias):
        x2, w1, w2, h, d = ctx.saved_tensors
        if gy is None:
            return g_alias, None, None, None, None
        g2 = gy.reshape(-1, gy.shape[-1]).contiguous()
        M, Fd = h.shape
        gw2 = _gemm.gemm_tn(g2, h) if ctx.needs_input_grad[3] else None
        gb2 = _bias_grad(gy, g2, ctx.bias_dtypes[1]) if ctx.needs_input_grad[4] else None
        # (g2 W2) * gelu'(pre); its column sums (the first layer's bias gradient) come out of the same store phase
        gb1 = None
        if ctx.needs_input_grad[2] and _fused_bias_grad:
            g1, gb1 = _gemm.gemm256(_gemm.NN, g2, w2, M, Fd, g2.shape[1], mul=d, col_sum=True)
            gb1 = gb1.to(ctx.bias_dtypes[0])
        else:
            g1 = _gemm.gemm256(_gemm.NN, g2, w2, M, Fd, g2.shape[1], mul=d)
            if ctx.needs_input_grad[2]:
                gb1 = _gemm.col_sum(g1, dtype=ctx.bias_dtypes[0])
        gw1 = _gemm.gemm_tn(g1, x2) if ctx.needs_input_grad[1] else None
        gx = None
        if ctx.needs_input_grad[0]:
            add = None
            if g_alias is not None and g_alias.dtype == torch.bfloat16 and g_alias.is_contiguous() \
                    and g_alias.data_ptr() % 16 == 0:
                add, g_alias = g_alias.reshape(-1, g_alias.shape[-1]), None
            gx = _gemm.gemm_nn(g1, w1, add=add).reshape(ctx.x_shape)
            if g_alias is not None:
                gx = gx + g_alias
        return gx, gw1, gb1, gw2, gb2


def ffn_fork(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor):
    """``(linear(gelu(linear(x, w1, b1)), w2, b2), x')`` -- the transformer feed-forward block with ``x'`` an
    alias of ``x`` for the residual branch. Large bf16 GPU shapes run as one autograd node with the GELU and
    its derivative inside the GEMM epilogues; anything else is the composition of :func:`linear_fork` and
    :func:`linear`."""
    M = x.numel() // x.shape[-1]
    H, Fd = x.shape[-1], w1.shape[0]
    if (x.is_cuda and x.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16
            and b1 is not None and b2 is not None and x.requires_grad and torch.is_grad_enabled()
            and _gemm.supported(x, w1) and w2.shape == (H, Fd) and w2.is_contiguous() and w1.is_contiguous()
            and _gemm.big_ok(M, Fd, H) and _gemm.big_ok(M, Fd, w2.shape[0])):
        return _FfnForkFn.apply(x, w1, b1, w2, b2)
    f, xr = linear_fork(x, w1, b1, "gelu")
    return linear(f, w2, b2), xr


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None,
           activation: str | None = None, dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    """``dropout(act(x @ weight.T + bias))`` with ``weight`` in torch ``(out, in)`` layout. On the bf16
    GPU path the dropout mask comes from the Philox state of ``ops/rng.py`` and is applied in the
    GEMM epilogue; elsewhere ``torch.nn.functional.dropout`` is used."""
    p = float(dropout_p) if training else 0.0
    if x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and _gemm.supported(x, weight):
        return _LinearFn.apply(x, weight, bias, activation, p)
    if (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and activation is None and p == 0.0
            and weight.shape[0] % 8 and weight.shape[0] >= 64 and _gemm.supported(x, weight, any_width=True)):
        return _LinearPadNFn.apply(x, weight, bias)
    if x.is_cuda and weight.is_cuda and x.numel() > 0 and x.dtype in (torch.float32, torch.bfloat16) \
            and weight.dtype in (torch.float32, torch.bfloat16):
        return _LinearAnyFn.apply(x, weight, bias, activation, p)
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


class _DropoutFn(torch.autograd.Function):
    """Stand-alone dropout on the GPU: Philox mask from ``(seed, device-resident offset)``, regenerated in backward."""

    @staticmethod
    def forward(ctx, x, p):
        st = _rng.state_for(x.device)
        ctx.drop = (p, st.seed, st.snapshot())
        y = _any.ew_f32(3, x, p=p, seed=st.seed, off=ctx.drop[2]) if x.dtype == torch.float32 \
            else _gemm.dropout_fwd(x, *ctx.drop)
        st.advance(x.numel())
        return y.view(x.shape) if y.shape != x.shape else y

    @staticmethod
    def backward(ctx, g):
        p, seed, off = ctx.drop
        out = _any.ew_f32(3, g, p=p, seed=seed, off=off) if g.dtype == torch.float32 \
            else _gemm.dropout_bwd(g, p, seed, off)
        return out.view(g.shape), None


def dropout(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    """Inverted dropout. GPU fp32 / bf16 tensors draw their mask from the framework's Philox state in memory order
    (a channels_last activation keeps its layout); CPU tensors use torch's generator."""
    if not training or p <= 0.0:
        return x
    if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.numel() > 0:
        if x.is_contiguous():
            return _DropoutFn.apply(x, float(p))
        if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
            return _DropoutFn.apply(x.permute(0, 2, 3, 1), float(p)).permute(0, 3, 1, 2)
        return _DropoutFn.apply(x.contiguous(), float(p))
    return F.dropout(x, p, True)
