"""Front end of the general-purpose MFMA GEMM (``csrc/kernels/gemm_any.hip``): exact-fp32 products of fp32 or bf16
operands with any strides and sizes, plus im2col / col2im and the fp32 elementwise kernels that go with it.

It carries what the bf16 tile kernels do not take on a GPU: fp32 models (the reference's configuration: reference
initializer.py:40, centralized/client.py:80-81) and bf16 products with rows that are not 16-byte aligned (the
512 -> 10 output layer, reference initializer.py:19), so that no dense or convolution layer of a GPU run falls
through to a vendor BLAS / convolution library.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

_ACT = {None: 0, "relu": 1, "gelu": 2}
_N_CU = 256
_ws: dict = {}


def _workspace(device, numel: int) -> torch.Tensor:
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _ws.get(key)
    if t is None or t.numel() < numel:
        t = _ws[key] = torch.empty(max(numel, 1 << 20), dtype=torch.float32, device=device)
    return t


def usable(*ts: torch.Tensor) -> bool:
    return all(t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) for t in ts)


def _pick_splits(M: int, N: int, K: int) -> int:
    # few output tiles: the product is one block's serial k-loop long unless the contraction is cut so that about
    # one block per CU runs (the layers of the reference MLP at batch 100 are 16 tiles of 49 k-steps each)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    if 2 * tiles > _N_CU or K < 128:
        return 1
    return max(1, min(_N_CU // tiles, K // 64, 256))


def matmul(a: torch.Tensor, b: torch.Tensor, bias: torch.Tensor | None = None, activation: str | None = None,
           out_dtype: torch.dtype | None = None, out: torch.Tensor | None = None, splits: int | None = None):
    """``act(a @ b + bias)`` for 2-D views ``a [M, K]`` and ``b [K, N]`` with ANY strides (pass ``w.t()`` for a
    ``[N, K]`` weight, ``x.t()`` for a weight gradient): no copy is made, the kernel reads through the strides."""
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[0] or not usable(a, b):
        raise ValueError("gemm_any.matmul: needs GPU fp32 / bf16 matrices [M, K] and [K, N]")
    M, K = a.shape
    N = b.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or (torch.float32 if torch.float32 in (a.dtype, b.dtype)
                                                      else torch.bfloat16), device=a.device)
    if out.stride(1) != 1 and N > 1:
        raise ValueError("gemm_any.matmul: the output must be row-major")
    if bias is not None and (bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("gemm_any.matmul: bias must be a contiguous [N] vector")
    if splits is None:
        splits = _pick_splits(M, N, K)
    ws = _workspace(a.device, splits * M * N) if splits > 1 else None
    rc = _lib.kernels().dtf_gemm_any(
        _lib.ptr(a), _lib.dtype_code(a.dtype), ctypes.c_longlong(a.stride(0)), ctypes.c_longlong(a.stride(1)),
        _lib.ptr(b), _lib.dtype_code(b.dtype), ctypes.c_longlong(b.stride(0)), ctypes.c_longlong(b.stride(1)),
        _lib.ptr(out), _lib.dtype_code(out.dtype), ctypes.c_longlong(out.stride(0) if M > 1 else N), _lib.ptr(bias),
        _lib.dtype_code(bias.dtype) if bias is not None else 0, _ACT[activation], M, N, K, splits, _lib.ptr(ws),
        _lib.stream_ptr(a.device))
    _lib.check(rc, "gemm_any")
    return out


def conv_product(mode: str, img: torch.Tensor, mat: torch.Tensor, geom: tuple, M: int, N: int, K: int,
                 bias: torch.Tensor | None = None, activation: str | None = None,
                 out_dtype: torch.dtype | None = None) -> torch.Tensor:
    """One pass of a convolution as an IMPLICIT product: the patch matrix of the contiguous NHWC tensor ``img`` is
    gathered while the kernel stages it, never written to memory.

    ``mode='fwd'``:   ``[M = N*OH*OW, N = K_out] = patches(x) [M, K = KH*KW*C] @ mat [K, N]`` (``mat`` any strides)
    ``mode='wgrad'``: ``[M = K_out, N = KH*KW*C] = mat [M, K = N*OH*OW] @ patches(x) [K, N]``
    ``mode='dgrad'``: ``[M = N*H*W, N = C_in] = patches'(dy) [M, K = KH*KW*K_out] @ mat [K, N]`` with the transposed
    patch geometry (``img`` = dy; ``mat`` = the weight as ``[(kh, kw, k_out), c_in]``)
    ``geom`` = (H, W, C, OH, OW, KH, KW, SH, SW, PT, PL): the image's own size and channels first, then the other side's
    spatial size (for 'dgrad' the image is dy, so H, W, C = OH, OW, K_out and OH, OW = the input's H, W)."""
    gm = {"fwd": 1, "wgrad": 2, "dgrad": 3}[mode]
    if not usable(img, mat) or img.dtype != mat.dtype or not img.is_contiguous() or mat.dim() != 2:
        raise ValueError("gemm_any.conv_product: needs a contiguous NHWC image and a matrix of the same GPU float type")
    out = torch.empty((M, N), dtype=out_dtype or img.dtype, device=img.device)
    if bias is not None and (bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("gemm_any.conv_product: bias must be a contiguous [N] vector")
    splits = _pick_splits(M, N, K)
    ws = _workspace(img.device, splits * M * N) if splits > 1 else None
    g = (ctypes.c_int * 11)(*[int(v) for v in geom])
    rc = _lib.kernels().dtf_gemm_any_conv(
        gm, _lib.ptr(img), _lib.ptr(mat), _lib.dtype_code(img.dtype), ctypes.c_longlong(mat.stride(0)),
        ctypes.c_longlong(mat.stride(1)), g, _lib.ptr(out), _lib.dtype_code(out.dtype), ctypes.c_longlong(N),
        _lib.ptr(bias), _lib.dtype_code(bias.dtype) if bias is not None else 0, _ACT[activation], M, N, K, splits,
        _lib.ptr(ws), _lib.stream_ptr(img.device))
    _lib.check(rc, "gemm_any_conv")
    return out


def im2col(x_nhwc: torch.Tensor, KH, KW, SH, SW, PT, PL, OH, OW) -> torch.Tensor:
    """``[N*OH*OW, KH*KW*C]`` patch matrix of a contiguous ``[N, H, W, C]`` image (zero padded)."""
    N, H, W, C = x_nhwc.shape
    cols = torch.empty((N * OH * OW, KH * KW * C), dtype=x_nhwc.dtype, device=x_nhwc.device)
    rc = _lib.kernels().dtf_im2col(0, _lib.dtype_code(x_nhwc.dtype), _lib.ptr(x_nhwc), _lib.ptr(cols), N, H, W, C, OH,
                                   OW, KH, KW, SH, SW, PT, PL, _lib.stream_ptr(x_nhwc.device))
    _lib.check(rc, "im2col")
    return cols


def col2im(cols: torch.Tensor, shape_nhwc, KH, KW, SH, SW, PT, PL, OH, OW) -> torch.Tensor:
    """Transpose of :func:`im2col`: every image element sums the patch entries that were copied from it."""
    N, H, W, C = shape_nhwc
    x = torch.empty((N, H, W, C), dtype=cols.dtype, device=cols.device)
    rc = _lib.kernels().dtf_im2col(1, _lib.dtype_code(cols.dtype), _lib.ptr(cols), _lib.ptr(x), N, H, W, C, OH, OW, KH,
                                   KW, SH, SW, PT, PL, _lib.stream_ptr(cols.device))
    _lib.check(rc, "col2im")
    return x


def ew_f32(op: int, a: torch.Tensor, b: torch.Tensor | None = None, p: float = 0.0, scale: float = 1.0,
           seed: int = 0, off: torch.Tensor | None = None) -> torch.Tensor:
    """fp32 elementwise kernel: op 0 ``a * (b > 0)``, 1 ``a * gelu'(b)``, 2 ``gelu(a)``, 3 ``a``; ``p`` > 0 applies
    Philox dropout (mask from ``(seed, off)``) to ``a`` first."""
    a = a.contiguous()
    b = b.contiguous() if b is not None else None
    out = torch.empty_like(a)
    rc = _lib.kernels().dtf_ew_f32(op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), ctypes.c_longlong(a.numel()),
                                   ctypes.c_float(p), ctypes.c_float(scale), ctypes.c_ulonglong(seed), _lib.ptr(off),
                                   _lib.stream_ptr(a.device))
    _lib.check(rc, "ew_f32")
    return out
