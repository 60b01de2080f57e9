"""Python front end of the bf16 MFMA GEMM (``csrc/kernels/gemm.hip``).

Three entry points, one per product of a dense / 1x1-convolution layer:

* :func:`gemm_nt` -- ``C[M,N] = act(A[M,K] @ B[N,K]^T + bias)`` (forward);
* :func:`gemm_nn` -- ``C[M,N] = A[M,K] @ B[K,N]`` (data gradient);
* :func:`gemm_tn` -- ``C[M,N] = A[K,M]^T @ B[K,N]`` (weight gradient; the contraction
  runs over the batch, so it is split-K'd to fill the 256 CUs).

The kernel needs 16-byte aligned rows (leading dimensions and the contiguous
extents multiples of 8 bf16); :func:`supported` reports whether a pair of
operands qualifies. Callers route everything else to the library GEMM.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib

_ACT = {None: 0, "relu": 1, "gelu": 2}
NT, NN, TN = 0, 1, 2
_N_CU = 256
_SPLIT_BLOCKS_WIDE = int(os.environ.get("DTF_GEMM_SPLIT_BLOCKS_WIDE", "4"))       # ... and for wider ones
_SPLIT_BLOCKS_PER_CU = int(os.environ.get("DTF_GEMM_SPLIT_BLOCKS_PER_CU", "4"))   # split-K target for products of <= 16 tiles
# (with the blocks of a split sharing an XCD: 6 / 4 measured 76.0 ms per ResNet-50 step, 4 / 3 77.0, 8 / 4 75.9, 12 / 6 76.5;
# re-measured with the weight gradients on the side stream and the leaner norm kernels, same box, images/s: 4 / 4 14,028-14,048,
# 3 / 4 14,035, 6 / 4 13,985-14,010, 8 / 4 13,939, 4 / 5 13,935, 6 / 6 13,920 -- fewer fp32 slabs to write and re-read)


def _ok2d(t: torch.Tensor) -> bool:
    return (t.dim() == 2 and t.dtype == torch.bfloat16 and t.stride(1) == 1 and t.stride(0) % 8 == 0
            and t.data_ptr() % 16 == 0 and t.shape[1] % 8 == 0)


def supported(x: torch.Tensor, w: torch.Tensor, any_width: bool = False) -> bool:
    """True when ``x[..., K] @ w[N, K]^T`` and both of its gradients can run on the MFMA kernel. ``any_width``: N need
    not be a multiple of 8 (the caller keeps the output in a buffer with padded rows, ``ops.dense._LinearPadNFn``)."""
    K = x.shape[-1]
    N = w.shape[0]
    return (x.is_cuda and K % 8 == 0 and (N % 8 == 0 or any_width) and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
            and w.is_contiguous() and x.numel() > 0
            # operands are addressed with 32-bit byte offsets (buffer descriptors): below 2^31 elements each
            and x.numel() < (1 << 31) and w.numel() < (1 << 31) and (x.numel() // K) * N < (1 << 31))


class GemmExtra(ctypes.Structure):
    """Mirror of the kernel's ``GemmExtra``: batch dims/strides, convolution geometry, accumulate input."""
    _fields_ = [("nb1", ctypes.c_int), ("nb2", ctypes.c_int),
                ("sA1", ctypes.c_longlong), ("sA2", ctypes.c_longlong), ("sB1", ctypes.c_longlong),
                ("sB2", ctypes.c_longlong), ("sC1", ctypes.c_longlong), ("sC2", ctypes.c_longlong),
                ("conv", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("OH", ctypes.c_int),
                ("OW", ctypes.c_int), ("S", ctypes.c_int), ("P", ctypes.c_int), ("KH", ctypes.c_int),
                ("KW", ctypes.c_int), ("CI", ctypes.c_int), ("CO", ctypes.c_int), ("NB", ctypes.c_int),
                ("Cin", ctypes.c_void_p),
                ("drop_p", ctypes.c_float), ("drop_pad", ctypes.c_int), ("drop_seed", ctypes.c_ulonglong),
                ("drop_off", ctypes.c_void_p), ("col_stats", ctypes.c_void_p),
                ("stats_mode", ctypes.c_int), ("bn_relu", ctypes.c_int), ("bn_x", ctypes.c_void_p),
                ("bn_mean", ctypes.c_void_p), ("bn_scale", ctypes.c_void_p), ("bn_shift", ctypes.c_void_p),
                ("bn_mask", ctypes.c_void_p), ("cin_mask", ctypes.c_void_p),
                ("SW", ctypes.c_int), ("PW", ctypes.c_int), ("use_sw", ctypes.c_int), ("fold", ctypes.c_int),
                ("tile", ctypes.c_int), ("up2", ctypes.c_int),
                ("tf_scale", ctypes.c_void_p), ("tf_shift", ctypes.c_void_p), ("tf_relu", ctypes.c_int)]


CONV_FWD, CONV_DGRAD, CONV_WGRAD, CONV_WGRAD_T = 1, 2, 3, 4


def conv_geom(mode: int, H: int, W: int, OH: int, OW: int, stride: int, pad: int, KH: int, KW: int, CI: int,
              CO: int, NB: int = 1, stride_w: int | None = None, pad_w: int | None = None,
              fold: bool = False) -> GemmExtra:
    """Geometry of an NHWC convolution for the implicit-GEMM modes of the kernel (``NB`` images).
    ``stride_w`` / ``pad_w``: stride and left padding along W when they differ from those along H (forward and
    weight gradient). ``fold`` (weight gradient): one launch plane per kernel row with its ``KW`` taps side by
    side in the output columns (``N = KW * CI``) instead of one plane per tap -- for narrow ``CI``.
    ``CONV_WGRAD_T`` is the transposed weight gradient ``[KH*KW*CI, CO] = im2col(x)^T @ dy`` in one plane."""
    ex = GemmExtra()
    ex.nb1 = ex.nb2 = 1
    ex.conv = mode
    ex.NB = NB
    ex.H, ex.W, ex.OH, ex.OW, ex.S, ex.P, ex.KH, ex.KW, ex.CI, ex.CO = H, W, OH, OW, stride, pad, KH, KW, CI, CO
    if stride_w is not None or pad_w is not None:
        ex.use_sw = 1
        ex.SW = stride if stride_w is None else stride_w
        ex.PW = pad if pad_w is None else pad_w
    if mode == CONV_WGRAD:
        if fold:                 # plane kh writes columns [kh*KW*CI, (kh+1)*KW*CI) of dW
            ex.fold, ex.nb2, ex.sC2 = 1, KH, KW * CI
        else:                    # one launch plane per filter tap; tap t writes columns [t*CI, (t+1)*CI) of dW
            ex.nb2 = KH * KW
            ex.sC2 = CI
    return ex


def batch_dims(nb1: int, nb2: int, sA, sB, sC) -> GemmExtra:
    """Two-level batch: operand offsets ``b1 * s[0] + b2 * s[1]`` elements."""
    ex = GemmExtra()
    ex.nb1, ex.nb2 = nb1, nb2
    ex.sA1, ex.sA2 = sA
    ex.sB1, ex.sB2 = sB
    ex.sC1, ex.sC2 = sC
    return ex


_workspaces: dict[tuple, torch.Tensor] = {}


def _workspace(device, numel: int) -> torch.Tensor:
    # one workspace per stream: launches on one stream are ordered, launches on different streams
    # (parameter-server service threads) must not share scratch memory
    key = (device.type, device.index, _lib.stream_ptr(device).value if device.type == "cuda" else 0)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < numel:
        ws = torch.empty(max(numel, 1 << 22), dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def tile_variant(N: int) -> int:
    """Output tile of a product ``N`` columns wide: 1 = 128 x 64 (at most 64 columns: half the accumulators and
    B tile, four resident blocks per CU), 0 = 128 x 128."""
    return 1 if (_narrow_tile and N <= 64) else 0


_narrow_tile = os.environ.get("DTF_GEMM_TILE64", "1") == "1"


def _pick_splits(M: int, N: int, K: int) -> int:
    """Split-K factor: enough blocks for 3-4 per CU, at least 8 k-tiles (512 deep) per split."""
    bn = 64 if tile_variant(N) else 128
    tiles = ((M + 127) // 128) * ((N + bn - 1) // bn)
    if tiles >= _N_CU or K <= 1024:
        return 1
    # few output tiles and a very long contraction (convolution weight gradients over a large batch) are pure
    # streaming: one more resident block per CU hides more latency; wider outputs measured better with 3
    per_cu = _SPLIT_BLOCKS_PER_CU if tiles <= 16 else _SPLIT_BLOCKS_WIDE
    want = -(-per_cu * _N_CU // tiles)
    return max(1, min(want, K // 512, 512))


def _launch(layout: int, A, B, M, N, K, lda, ldb, bias, activation, out_dtype, splits, out=None, extra=None,
            ldc=None):
    dev = A.device
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=dev)
    if ldc is None:
        ldc = out.stride(0)
        if out.stride(1) != 1:
            raise ValueError("gemm output must be row-major")
    if ldc % 8 or ldc < (N + 7) // 8 * 8 or out.data_ptr() % 16:
        raise ValueError("gemm output needs 16-byte aligned rows that hold N rounded up to a multiple of 8 elements")
    batch = 1
    if tile_variant(N):
        if extra is None:
            extra = GemmExtra()
            extra.nb1 = extra.nb2 = 1
        extra.tile = 1
    if extra is not None:
        batch = max(1, extra.nb1) * max(1, extra.nb2)
        if extra.up2 or extra.Cin or (batch > 1 and extra.conv != CONV_WGRAD) or (extra.conv == CONV_DGRAD and extra.S == 2):
            splits = 1
    if splits is None:
        splits = _pick_splits(M * batch if batch > 1 else M, N, K)
    ws = _workspace(dev, splits * batch * M * N) if splits > 1 else None
    bias_dtype = 0
    if bias is not None:
        if bias.numel() != N or not bias.is_contiguous():
            raise ValueError("bias must be a contiguous [N] vector")
        bias_dtype = _lib.dtype_code(bias.dtype)
    log = launch_log
    if log is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.kernels().dtf_gemm_bf16_ex(layout, _lib.ptr(A), _lib.ptr(B), _lib.ptr(out), _lib.ptr(bias), M, N, K,
                                         ctypes.c_longlong(lda), ctypes.c_longlong(ldb), ctypes.c_longlong(ldc),
                                         bias_dtype, _ACT[activation], 1 if out.dtype == torch.float32 else 0,
                                         splits, _lib.ptr(ws), ctypes.byref(extra) if extra is not None else None,
                                         _lib.stream_ptr(dev))
    _lib.check(rc, "gemm_bf16")
    if log is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        e = extra
        log.append((dict(layout=layout, M=M, N=N, K=K, batch=batch, splits=splits, fp32_out=out.dtype == torch.float32,
                         conv=e.conv if e else 0, H=e.H if e else 0, W=e.W if e else 0, OH=e.OH if e else 0,
                         OW=e.OW if e else 0, S=e.S if e else 0, KH=e.KH if e else 0, KW=e.KW if e else 0,
                         CI=e.CI if e else 0, CO=e.CO if e else 0, NB=e.NB if e else 0, add=bool(e and e.Cin),
                         add_mask=bool(e and e.cin_mask), stats=bool(e and e.col_stats),
                         bn_grad=bool(e and e.stats_mode), bn_mask=bool(e and e.bn_mask), in_bn=bool(e and e.tf_scale),
                         tile=e.tile if e else 0, up2=e.up2 if e else 0, bias=bias is not None,
                         act=activation), ev0, ev1))
    return out


# tools/roofline_step.py sets this to a list: every launch of the 128-wide kernel family then appends
# (description, start event, end event) on the current stream (None: no logging, no cost)
launch_log: list | None = None


_big_tile = os.environ.get("DTF_GEMM256", "1") == "1"


def set_big_tile(on: bool) -> None:
    """Routes large NT products to the 256x256 deep-pipelined kernel (A/B measurements switch it off)."""
    global _big_tile
    _big_tile = bool(on)


def _big_ok(M: int, N: int, K: int) -> bool:
    # compute-bound shapes only: the 256x256 tile needs >= ~1 tile per CU and a k-loop worth pipelining
    if not (_big_tile and K % 64 == 0 and K >= 256 and N >= 256 and M >= 512):
        return False
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    waves = -(-tiles // _N_CU)
    return tiles / (waves * _N_CU) >= 0.7          # one tile per CU at a time: a ragged last wave idles CUs


def gemm256(layout: int, a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, bias=None, activation=None,
            out=None, splits: int = 1, add: torch.Tensor | None = None, mul: torch.Tensor | None = None,
            grad_out: torch.Tensor | None = None, col_sum: bool = False):
    """``act(op(a) @ op(b) + bias)`` on the 256x256x64 LDS-DMA kernel (``csrc/kernels/gemm256.hip``);
    layouts as in this module (NT, NN, TN). ``splits`` > 1 splits the contraction (weight gradients).
    ``add`` / ``mul``: bf16 tensor laid out like the result, added to / multiplied into it in the epilogue.
    ``grad_out`` (with ``activation="gelu"``): receives ``gelu'(pre-activation)``.
    ``col_sum=True`` (with ``mul``): returns ``(result, fp32 column sums of the result)``; the store phase sums the
    rounded values it writes per wave and a finishing kernel adds the partial rows (the bias gradient of the layer whose
    output gradient the product is, without a pass over the result)."""
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    bias_dtype = _lib.dtype_code(bias.dtype) if bias is not None else 0
    ws = _workspace(a.device, splits * M * N) if splits > 1 else None
    act = _ACT[activation]
    if grad_out is not None:
        if activation != "gelu" or grad_out.shape != out.shape or grad_out.stride() != out.stride() \
                or grad_out.dtype != torch.bfloat16:
            raise ValueError("gemm256: grad_out needs activation='gelu' and the layout of the output")
        act = 3
    if add is not None and mul is not None:
        raise ValueError("gemm256: add= and mul= are exclusive")
    cin = add if add is not None else mul
    parts = None
    if col_sum:
        if mul is None or splits != 1:
            raise ValueError("gemm256: col_sum=True needs mul= and no split-K")
        parts = torch.empty((4 * ((M + 255) // 256), N), dtype=torch.float32, device=a.device)
        if M % 256:
            parts.zero_()                 # waves whose rows lie past M write nothing
    rc = _lib.kernels().dtf_gemm256_cs(layout, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), _lib.ptr(bias), M, N, K,
                                       ctypes.c_longlong(a.stride(0)), ctypes.c_longlong(b.stride(0)),
                                       ctypes.c_longlong(out.stride(0)), bias_dtype, act, splits,
                                       _lib.ptr(ws), _lib.ptr(cin), 2 if mul is not None else 1, _lib.ptr(grad_out),
                                       _lib.ptr(parts), _lib.stream_ptr(a.device))
    _lib.check(rc, "gemm256")
    if col_sum:
        sums = torch.empty(N, dtype=torch.float32, device=a.device)
        _lib.check(_lib.kernels().dtf_col_sum_finish(_lib.ptr(parts), _lib.ptr(sums), int(parts.shape[0]), N,
                                                     _lib.stream_ptr(a.device)), "col_sum_finish")
        return out, sums
    return out


def big_ok(M: int, N: int, K: int) -> bool:
    """True when an ``[M, K] x [K, N]`` product is routed to the 256x256 kernel (and its fused epilogues exist)."""
    return _big_ok(M, N, K)


def gemm256_nt(a: torch.Tensor, b: torch.Tensor, bias=None, activation=None, out=None) -> torch.Tensor:
    return gemm256(NT, a, b, a.shape[0], b.shape[0], a.shape[1], bias, activation, out)


def _big_splits(M: int, N: int, K: int) -> int:
    """Split-K factor of a weight-gradient shaped product on the 256x256 kernel (0: do not use it)."""
    if not (_big_tile and K % 64 == 0 and M >= 256 and N >= 256 and M % 8 == 0):
        return 0
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles >= _N_CU:
        waves = -(-tiles // _N_CU)
        return 1 if tiles / (waves * _N_CU) >= 0.7 and K >= 256 else 0
    splits = min(max(1, _N_CU // tiles), K // 512)     # one full wave of tiles: fewer fp32 slabs to reduce
    # few tiles need many splits: worth it while the fp32 slabs (written, then read by the reduction) stay below the
    # operands' bytes -- 768 x 768 over 65536 rows: 28 splits, 108 us against 133 us on the 128 x 128 kernel
    if tiles < 24 and splits * M * N * 4 > K * (M + N) * 2:
        return 0
    return splits if splits >= 1 and tiles * splits >= _N_CU // 2 else 0


def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias=None, activation=None, out_dtype=torch.bfloat16, splits=None,
            out=None, dropout=None) -> torch.Tensor:
    """``dropout(act(a[M,K] @ b[N,K]^T + bias))``. ``dropout = (p, seed, offset_tensor)`` applies Philox
    dropout in the epilogue (``offset_tensor``: int64 device scalar, see ``ops/rng.py``)."""
    if not (_ok2d(a) and _ok2d(b)) or a.shape[1] != b.shape[1]:
        raise ValueError("gemm_nt: operands must be 16-byte aligned bf16 [M,K] and [N,K]")
    extra = None
    if dropout is not None and dropout[0] > 0.0:
        if out_dtype != torch.bfloat16 or (out is not None and out.stride(0) != b.shape[0]):
            raise ValueError("gemm_nt: fused dropout needs a dense bf16 output")
        extra = GemmExtra()
        extra.nb1 = extra.nb2 = 1
        extra.drop_p, extra.drop_seed, extra.drop_off = float(dropout[0]), int(dropout[1]), dropout[2].data_ptr()
        splits = 1
    elif (out_dtype == torch.bfloat16 and splits is None and _big_ok(a.shape[0], b.shape[0], a.shape[1])
          and (out is None or (out.dtype == torch.bfloat16 and out.stride(1) == 1 and out.stride(0) % 8 == 0))
          and (bias is None or (bias.numel() == b.shape[0] and bias.is_contiguous()))):
        return gemm256_nt(a, b, bias, activation, out)
    return _launch(NT, a, b, a.shape[0], b.shape[0], a.shape[1], a.stride(0), b.stride(0), bias, activation,
                   out_dtype, splits, out, extra=extra)


def gemm_nn(a: torch.Tensor, b: torch.Tensor, out_dtype=torch.bfloat16, splits=None, out=None,
            add: torch.Tensor | None = None) -> torch.Tensor:
    """``a[M,K] @ b[K,N] (+ add)``; ``add`` is a contiguous bf16 ``[M, N]`` tensor accumulated in the epilogue."""
    if not (_ok2d(a) and _ok2d(b)) or a.shape[1] != b.shape[0]:
        raise ValueError("gemm_nn: operands must be 16-byte aligned bf16 [M,K] and [K,N]")
    if add is not None:
        if (add.dtype != torch.bfloat16 or tuple(add.shape) != (a.shape[0], b.shape[1]) or not add.is_contiguous()
                or add.data_ptr() % 16 or out_dtype != torch.bfloat16 or out is not None):
            raise ValueError("gemm_nn: add= needs a contiguous 16-byte aligned bf16 [M, N] tensor")
    if (out_dtype == torch.bfloat16 and splits is None and _big_ok(a.shape[0], b.shape[1], a.shape[1])
            and (out is None or (out.dtype == torch.bfloat16 and out.stride(1) == 1 and out.stride(0) % 8 == 0))):
        return gemm256(NN, a, b, a.shape[0], b.shape[1], a.shape[1], out=out, add=add)
    if add is not None:
        M, N, K = a.shape[0], b.shape[1], a.shape[1]
        res = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
        return gemm_raw(NN, a, b, res, M, N, K, a.stride(0), b.stride(0), N, add=add)
    return _launch(NN, a, b, a.shape[0], b.shape[1], a.shape[1], a.stride(0), b.stride(0), None, None, out_dtype,
                   splits, out)


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out_dtype=torch.bfloat16, splits=None, out=None) -> torch.Tensor:
    """``a[K,M]^T @ b[K,N]``."""
    if not (_ok2d(a) and _ok2d(b)) or a.shape[0] != b.shape[0]:
        raise ValueError("gemm_tn: operands must be 16-byte aligned bf16 [K,M] and [K,N]")
    if out_dtype == torch.bfloat16 and splits is None and (out is None or (out.dtype == torch.bfloat16 and
                                                                          out.stride(1) == 1 and out.stride(0) % 8 == 0)):
        sp = _big_splits(a.shape[1], b.shape[1], a.shape[0])
        if sp:
            return gemm256(TN, a, b, a.shape[1], b.shape[1], a.shape[0], out=out, splits=sp)
    return _launch(TN, a, b, a.shape[1], b.shape[1], a.shape[0], a.stride(0), b.stride(0), None, None, out_dtype,
                   splits, out)


def col_stats_rows(M: int, N: int) -> int:
    """Rows of the column-statistics buffer of an ``M x N`` product: one per 128-row output tile (the waves of a
    block combine their sums before they are written)."""
    return (M + 127) // 128


def gemm_raw(layout: int, A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int, lda: int,
             ldb: int, ldc: int, extra: GemmExtra | None = None, bias=None, activation=None, splits=None,
             add: torch.Tensor | None = None, col_stats: torch.Tensor | None = None, bn_grad=None,
             add_mask: torch.Tensor | None = None, in_bn=None):
    """Unchecked launch on explicit dimensions and leading dimensions (batched attention products,
    implicit-GEMM convolutions). ``add`` is a bf16 tensor laid out like ``out`` that is added to the
    result in the epilogue. ``col_stats`` (fp32 ``[col_stats_rows(M, N), 2, N]``)
    receives per-tile column sums and sums of squares of the bf16 output, the partial statistics of a
    batch norm that follows. ``in_bn = (scale, shift, relu)`` (fp32 per-channel vectors): the image operand (A of a
    forward product, the activation operand of a weight gradient) is the INPUT of a batch norm and is normalised
    (and rectified) while it is staged, see ``ops.norm.batch_norm_deferred``.
    The caller guarantees 16-byte alignment of every row it addresses."""
    if in_bn is not None:
        sc, sh, relu = in_bn
        if sc.dtype != torch.float32 or sh.dtype != torch.float32 or not (sc.is_contiguous() and sh.is_contiguous()):
            raise ValueError("gemm_raw: in_bn scale / shift must be contiguous fp32 vectors")
        if extra is None:
            extra = GemmExtra()
            extra.nb1 = extra.nb2 = 1
        extra.tf_scale, extra.tf_shift, extra.tf_relu = sc.data_ptr(), sh.data_ptr(), int(bool(relu))
    if col_stats is not None:
        if (col_stats.dtype != torch.float32 or not col_stats.is_contiguous() or out.dtype != torch.bfloat16
                or tuple(col_stats.shape) != (col_stats_rows(M, N), 2, N)):
            raise ValueError("gemm_raw: col_stats must be a contiguous fp32 [col_stats_rows(M, N), 2, N] tensor")
        if extra is None:
            extra = GemmExtra()
            extra.nb1 = extra.nb2 = 1
        extra.col_stats = col_stats.data_ptr()
        if bn_grad is not None:
            # the output is the gradient entering a batch norm: bn_grad = (norm input laid out like out,
            # stats [4, N] = mean, invstd, scale, shift, mask bytes or None, relu code 0 | 2 | 3)
            bx, bstats, bmask, brelu = bn_grad
            if bx.dtype != torch.bfloat16 or bx.numel() != M * N or ldc != N or bstats.dtype != torch.float32 \
                    or tuple(bstats.shape) != (4, N) or (brelu == 3 and bmask is None):
                raise ValueError("gemm_raw: bn_grad does not match the output")
            extra.stats_mode, extra.bn_relu = 1, int(brelu)
            extra.bn_x, extra.bn_mean = bx.data_ptr(), bstats[0].data_ptr()
            extra.bn_scale, extra.bn_shift = bstats[2].data_ptr(), bstats[3].data_ptr()
            extra.bn_mask = bmask.data_ptr() if bmask is not None else None
    for t in (A, B, out):
        if t.data_ptr() % 16:
            raise ValueError("gemm_raw: operands must be 16-byte aligned")
    if add is not None:
        if add.data_ptr() % 16 or add.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
            raise ValueError("gemm_raw: add= needs 16-byte aligned bf16 tensors")
        if extra is None:
            extra = GemmExtra()
            extra.nb1 = extra.nb2 = 1
        extra.Cin = add.data_ptr()
        if add_mask is not None:      # uint8, one bit per element of ``add``: elements whose bit is 0 are not added
            if add_mask.dtype != torch.uint8 or add_mask.numel() * 8 != M * N or ldc != N:
                raise ValueError("gemm_raw: add_mask must hold one bit per element of a dense add= tensor")
            extra.cin_mask = add_mask.data_ptr()
    if lda % 8 or ldb % 8 or ldc % 8:
        raise ValueError("gemm_raw: leading dimensions must be multiples of 8 elements")
    return _launch(layout, A, B, M, N, K, lda, ldb, bias, activation, out.dtype, splits, out=out, extra=extra, ldc=ldc)


def _ew(op: int, a: torch.Tensor, b: torch.Tensor | None) -> torch.Tensor:
    a = a.contiguous()
    if b is not None:
        b = b.contiguous()
    out = torch.empty_like(a)
    rc = _lib.kernels().dtf_ew_bf16(op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), ctypes.c_longlong(a.numel()),
                                    _lib.stream_ptr(a.device))
    _lib.check(rc, "ew_bf16")
    return out


def _ew_drop(op: int, a: torch.Tensor, b: torch.Tensor | None, p: float, scale: float, seed: int,
             off: torch.Tensor | None) -> torch.Tensor:
    a = a.contiguous()
    if b is not None:
        b = b.contiguous()
    out = torch.empty_like(a)
    rc = _lib.kernels().dtf_ew_drop_bf16(op, _lib.ptr(a), _lib.ptr(b), _lib.ptr(out), ctypes.c_longlong(a.numel()),
                                         ctypes.c_float(p), ctypes.c_float(scale), ctypes.c_ulonglong(seed),
                                         _lib.ptr(off), _lib.stream_ptr(a.device))
    _lib.check(rc, "ew_drop_bf16")
    return out


def dropout_fwd(x: torch.Tensor, p: float, seed: int, off: torch.Tensor) -> torch.Tensor:
    """Philox dropout of a contiguous bf16 tensor (mask = f(seed, off, linear index))."""
    return _ew_drop(3, x, None, p, 1.0, seed, off)


def dropout_bwd(g: torch.Tensor, p: float, seed: int, off: torch.Tensor, activation: str | None = None,
                saved: torch.Tensor | None = None) -> torch.Tensor:
    """Gradient through ``dropout(act(z))``: the forward mask is regenerated from ``(seed, off)``.
    ``saved`` is the layer OUTPUT for relu and the PRE-activation for gelu."""
    if activation == "relu":      # y > 0 already implies "kept": only the 1/(1-p) scale is needed
        return _ew_drop(0, g, saved, 0.0, 1.0 / (1.0 - p), 0, None)
    if activation == "gelu":
        return _ew_drop(1, g, saved, p, 1.0, seed, off)
    return _ew_drop(3, g, None, p, 1.0, seed, off)


def relu_bwd(g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``g * (y > 0)`` from the ReLU OUTPUT."""
    if g.is_cuda and g.dtype == torch.bfloat16:
        return _ew(0, g, y)
    return g * (y > 0).to(g.dtype)


def gelu_bwd(g: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    """``g * gelu'(pre)`` (erf form) from the PRE-activation."""
    if g.is_cuda and g.dtype == torch.bfloat16:
        return _ew(1, g, pre)
    x = pre.float()
    cdf = 0.5 * (1 + torch.erf(x * 0.7071067811865476))
    pdf = torch.exp(-0.5 * x * x) * 0.3989422804014327
    return (g.float() * (cdf + x * pdf)).to(g.dtype)


def gelu_fwd(x: torch.Tensor) -> torch.Tensor:
    if x.is_cuda and x.dtype == torch.bfloat16:
        return _ew(2, x, None)
    return torch.nn.functional.gelu(x)


def col_sum(g: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """Column sums of a row-major [M, N] matrix (the bias gradient); deterministic two-stage reduction."""
    if g.is_cuda and g.dtype == torch.bfloat16 and g.shape[1] % 8 == 0 and g.stride(1) == 1 and g.data_ptr() % 16 == 0:
        lib = _lib.kernels()
        M, N = g.shape
        parts = lib.dtf_col_sum_parts(M, N)
        out = torch.empty(N, dtype=torch.float32, device=g.device)
        ws = _workspace(g.device, parts * N) if parts > 1 else None
        rc = lib.dtf_col_sum_bf16(_lib.ptr(g), _lib.ptr(out), M, N, ctypes.c_longlong(g.stride(0)), _lib.ptr(ws),
                                  _lib.stream_ptr(g.device))
        _lib.check(rc, "col_sum_bf16")
        return out.to(dtype)
    return g.float().sum(0).to(dtype)
