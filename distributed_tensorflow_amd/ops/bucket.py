"""Gradient-bucket operations: multi-tensor flatten / unflatten with fused scale
and cast, flat scale/cast, n-ary weighted sums and sum-of-squares.

These are the device-side replacement for the reference's per-step
``pickle.dumps(list_of_arrays)`` / ``pickle.loads`` + ``set_weights`` round trip
(reference centralized/client.py:85-90, centralized/server.py:104-105).
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import torch

from . import _lib


def _tables(tensors: Sequence[torch.Tensor], offsets: Sequence[int]):
    n = len(tensors)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    offs = (ctypes.c_longlong * n)(*[int(o) for o in offsets])
    nums = (ctypes.c_longlong * n)(*[t.numel() for t in tensors])
    return n, ptrs, offs, nums


def _check_list(tensors: Sequence[torch.Tensor], flat: torch.Tensor, offsets: Sequence[int]):
    if len(tensors) != len(offsets):
        raise ValueError("tensors and offsets differ in length")
    if not tensors:
        return None
    dt = tensors[0].dtype
    for t, o in zip(tensors, offsets):
        if t.dtype != dt:
            raise ValueError("all tensors of one flatten call must share a dtype")
        if not t.is_contiguous():
            raise ValueError("bucket tensors must be contiguous")
        if o < 0 or o + t.numel() > flat.numel():
            raise ValueError("tensor does not fit the flat buffer at its offset")
        if t.device != flat.device:
            raise ValueError("tensor and flat buffer are on different devices")
    return dt


def flatten(tensors: Sequence[torch.Tensor], flat: torch.Tensor, offsets: Sequence[int], scale: float = 1.0,
            scale_tensor: torch.Tensor | None = None) -> None:
    """flat[offsets[i] : offsets[i] + n_i] = scale * tensors[i], cast to flat.dtype."""
    dt = _check_list(tensors, flat, offsets)
    if dt is None:
        return
    if flat.is_cuda:
        n, ptrs, offs, nums = _tables(tensors, offsets)
        rc = _lib.kernels().dtf_bucket_flatten(n, ptrs, offs, nums, _lib.ptr(flat), _lib.dtype_code(dt),
                                               _lib.dtype_code(flat.dtype), _lib.ptr(scale_tensor),
                                               ctypes.c_float(scale), _lib.stream_ptr(flat.device))
        _lib.check(rc, "bucket_flatten")
        return
    s = scale * (float(scale_tensor) if scale_tensor is not None else 1.0)
    fv = flat.view(-1)
    for t, o in zip(tensors, offsets):
        fv[o:o + t.numel()].copy_(t.reshape(-1).float() * s if s != 1.0 else t.reshape(-1))


def unflatten(flat: torch.Tensor, tensors: Sequence[torch.Tensor], offsets: Sequence[int], scale: float = 1.0,
              scale_tensor: torch.Tensor | None = None) -> None:
    """tensors[i] = scale * flat[offsets[i] : offsets[i] + n_i], cast to the tensor dtype."""
    dt = _check_list(tensors, flat, offsets)
    if dt is None:
        return
    if flat.is_cuda:
        n, ptrs, offs, nums = _tables(tensors, offsets)
        rc = _lib.kernels().dtf_bucket_unflatten(n, ptrs, offs, nums, _lib.ptr(flat), _lib.dtype_code(dt),
                                                 _lib.dtype_code(flat.dtype), _lib.ptr(scale_tensor),
                                                 ctypes.c_float(scale), _lib.stream_ptr(flat.device))
        _lib.check(rc, "bucket_unflatten")
        return
    s = scale * (float(scale_tensor) if scale_tensor is not None else 1.0)
    fv = flat.view(-1)
    for t, o in zip(tensors, offsets):
        src = fv[o:o + t.numel()]
        t.view(-1).copy_(src.float() * s if s != 1.0 else src)


def scale_cast(src: torch.Tensor, dst: torch.Tensor, scale: float = 1.0,
               scale_tensor: torch.Tensor | None = None) -> None:
    """dst = scale * src over flat buffers (dtype conversion fp32 <-> bf16 fused)."""
    n = src.numel()
    if dst.numel() != n:
        raise ValueError("scale_cast: size mismatch")
    if src.is_cuda:
        if not (src.is_contiguous() and dst.is_contiguous()) or src.data_ptr() % 16 or dst.data_ptr() % 16:
            raise ValueError("scale_cast needs contiguous 16-byte aligned buffers")
        rc = _lib.kernels().dtf_flat_scale_cast(_lib.ptr(src), _lib.ptr(dst), ctypes.c_longlong(n),
                                                _lib.dtype_code(src.dtype), _lib.dtype_code(dst.dtype),
                                                _lib.ptr(scale_tensor), ctypes.c_float(scale),
                                                _lib.stream_ptr(src.device))
        _lib.check(rc, "flat_scale_cast")
        return
    s = scale * (float(scale_tensor) if scale_tensor is not None else 1.0)
    dst.view(-1).copy_(src.reshape(-1).float() * s)


def nary_axpy(dst: torch.Tensor, srcs: Sequence[torch.Tensor], weights: Sequence[float], w_dst: float = 1.0) -> None:
    """dst = w_dst * dst + sum_k weights[k] * srcs[k] (up to 8 sources, one pass)."""
    if len(srcs) != len(weights):
        raise ValueError("srcs and weights differ in length")
    n = dst.numel()
    for s in srcs:
        if s.numel() != n or s.dtype != dst.dtype or s.device != dst.device:
            raise ValueError("nary_axpy: sources must match dst in size, dtype and device")
    if dst.is_cuda:
        for t in (dst, *srcs):
            if not t.is_contiguous() or t.data_ptr() % 16:
                raise ValueError("nary_axpy needs contiguous 16-byte aligned buffers")
        first = True
        for i in range(0, max(len(srcs), 1), 8):
            chunk, w = srcs[i:i + 8], weights[i:i + 8]
            k = len(chunk)
            ptrs = (ctypes.c_void_p * max(k, 1))(*[t.data_ptr() for t in chunk])
            ws = (ctypes.c_float * max(k, 1))(*[float(x) for x in w])
            rc = _lib.kernels().dtf_nary_axpy(_lib.ptr(dst), ctypes.c_float(w_dst if first else 1.0), k, ptrs, ws,
                                              ctypes.c_longlong(n), _lib.dtype_code(dst.dtype),
                                              _lib.stream_ptr(dst.device))
            _lib.check(rc, "nary_axpy")
            first = False
        return
    acc = dst.view(-1).float() * w_dst
    for s, w in zip(srcs, weights):
        acc = acc + float(w) * s.reshape(-1).float()
    dst.view(-1).copy_(acc)


def sumsq(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """Sum of squares of a flat buffer as a 1-element fp32 tensor (accumulates into ``out`` if given)."""
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
    if x.is_cuda:
        if not x.is_contiguous() or x.data_ptr() % 16:
            raise ValueError("sumsq needs a contiguous 16-byte aligned buffer")
        rc = _lib.kernels().dtf_sumsq(_lib.ptr(x), ctypes.c_longlong(x.numel()), _lib.dtype_code(x.dtype),
                                      _lib.ptr(out), _lib.stream_ptr(x.device))
        _lib.check(rc, "sumsq")
    else:
        out += x.float().pow(2).sum()
    return out
