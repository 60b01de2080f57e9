"""Embedding lookup, dense and sparse gradient paths (front end of ``csrc/kernels/embedding.hip``).

* :func:`embedding` -- differentiable gather; the gradient is a dense fp32 scatter-add (float
  atomics). ``pos`` / ``typ`` tables fuse BERT's position and token-type embeddings into the same
  pass.
* :class:`ShardedEmbedding` -- a row-sharded table for the parameter-server strategies: rank r
  holds rows ``[r*V/W, (r+1)*V/W)``; lookups are an all-to-all of ids and rows, gradients are
  reduced per unique id (sort + segment sum) and applied to the touched rows only by a sparse
  optimizer kernel, so a step moves and updates O(batch) rows, never the whole table.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

SGD, ADAGRAD, ADAM = 0, 1, 2


def _gather(table, ids, pos=None, period=1, typ=None, tt=None, out_dtype=None, row_base=0):
    n = ids.numel()
    D = table.shape[1]
    out_dtype = out_dtype or table.dtype
    if table.is_cuda:
        out = torch.empty(n, D, dtype=out_dtype, device=table.device)
        ids = ids.reshape(-1).long().contiguous()
        if tt is not None:
            tt = tt.reshape(-1).long().contiguous()
        rc = _lib.kernels().dtf_emb_gather(_lib.ptr(table), _lib.ptr(ids), _lib.ptr(pos), int(period), _lib.ptr(typ),
                                           _lib.ptr(tt), _lib.ptr(out), ctypes.c_longlong(n), D,
                                           ctypes.c_longlong(row_base), ctypes.c_longlong(table.shape[0]),
                                           _lib.dtype_code(table.dtype), _lib.dtype_code(out_dtype),
                                           _lib.stream_ptr(table.device))
        _lib.check(rc, "emb_gather")
        return out
    loc = ids.reshape(-1).long() - row_base
    ok = (loc >= 0) & (loc < table.shape[0])
    out = table[loc.clamp(0, table.shape[0] - 1)].float() * ok[:, None]
    if pos is not None:
        out = out + pos[torch.arange(n, device=ids.device) % period].float()
    if typ is not None:
        out = out + typ[tt.reshape(-1).long()].float()
    return out.to(out_dtype)


def _scatter_add(g, ids, n_rows, row_base=0, period_rows=0):
    """fp32 [n_rows, D] with ``out[ids[i] - row_base] += g[i]`` (ids None: row = i % period_rows)."""
    D = g.shape[1]
    out = torch.zeros(n_rows, D, dtype=torch.float32, device=g.device)
    if g.is_cuda:
        g = g.contiguous()
        idp = ids.reshape(-1).long().contiguous() if ids is not None else None
        rc = _lib.kernels().dtf_emb_scatter_add(_lib.ptr(g), _lib.ptr(idp), _lib.ptr(out), ctypes.c_longlong(g.shape[0]),
                                                D, ctypes.c_longlong(row_base), ctypes.c_longlong(n_rows),
                                                int(period_rows), _lib.dtype_code(g.dtype), _lib.stream_ptr(g.device))
        _lib.check(rc, "emb_scatter_add")
        return out
    if ids is None:
        loc = torch.arange(g.shape[0], device=g.device) % period_rows
    else:
        loc = ids.reshape(-1).long() - row_base
    ok = (loc >= 0) & (loc < n_rows)
    out.index_add_(0, loc[ok], g.float()[ok])
    return out


def _table_grad(g, ids, n_rows, row_base=0):
    """fp32 [n_rows, D] gradient of a table from the row gradients ``g`` and their ``ids``: sort by id,
    sum each run of equal ids in registers, store once (GPU); ``index_add_`` elsewhere."""
    if not g.is_cuda:
        return _scatter_add(g, ids, n_rows, row_base)
    D = g.shape[1]
    g = g.contiguous()
    sorted_ids, perm = torch.sort(ids.reshape(-1).long(), stable=True)
    out = torch.zeros(n_rows, D, dtype=torch.float32, device=g.device)
    rc = _lib.kernels().dtf_emb_sorted_grad(_lib.ptr(g), _lib.ptr(sorted_ids), _lib.ptr(perm), _lib.ptr(out),
                                            ctypes.c_longlong(g.shape[0]), D, ctypes.c_longlong(row_base),
                                            ctypes.c_longlong(n_rows), _lib.dtype_code(g.dtype),
                                            _lib.stream_ptr(g.device))
    _lib.check(rc, "emb_sorted_grad")
    return out


# DTF_EMB_ONEHOT=0: gradients of tables of a few rows through the sorted-run kernel (A/B switch)
_onehot_small = __import__("os").environ.get("DTF_EMB_ONEHOT", "1") != "0"


class _EmbFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, pos, typ, tt):
        period = pos.shape[0] if pos is not None else 1
        if pos is not None:
            period = ids.shape[-1]
        out = _gather(table, ids, pos, period, typ, tt)
        ctx.save_for_backward(ids, tt if tt is not None else ids)
        ctx.meta = (table.shape[0], pos.shape[0] if pos is not None else 0, typ.shape[0] if typ is not None else 0,
                    period, table.dtype, tt is not None)
        return out.view(*ids.shape, table.shape[1])

    @staticmethod
    def backward(ctx, g):
        ids, tt = ctx.saved_tensors
        V, P, T, period, dt, has_tt = ctx.meta
        g2 = g.reshape(-1, g.shape[-1])
        dtab = _table_grad(g2, ids, V).to(dt) if ctx.needs_input_grad[0] else None
        dpos = dtyp = None
        if P and ctx.needs_input_grad[2]:
            # position j of every sequence takes pos[j]: the gradient is a sum over the batch axis
            H = g.shape[-1]
            if g2.is_cuda and g2.dtype == torch.bfloat16 and g2.shape[0] % period == 0:
                from . import gemm as _gemm
                d = _gemm.col_sum(g2.contiguous().view(-1, period * H)).view(period, H)
            else:
                d = _scatter_add(g2, None, period, period_rows=period)
            dpos = torch.zeros(P, H, dtype=torch.float32, device=g.device)
            dpos[:period] = d
            dpos = dpos.to(dt)
        if T and ctx.needs_input_grad[3]:
            if (_onehot_small and g2.is_cuda and g2.dtype == torch.bfloat16 and T <= 64 and g2.shape[0] % 8 == 0
                    and g2.shape[1] % 8 == 0):
                # a table of a few rows (token types): every id repeats tens of thousands of times, so the sorted-run
                # kernel would walk runs of 128 rows thread by thread and combine 500 pieces per row with atomics
                # (~300 us for 65536 tokens). one_hot(ids)^T @ g is the same sum as a split-K product (fp32
                # accumulation of the bf16 rows, deterministic): ~20 us.
                from . import gemm as _gemm
                Tp = -(-T // 8) * 8
                oh = torch.zeros((g2.shape[0], Tp), dtype=torch.bfloat16, device=g2.device)
                oh.scatter_(1, tt.reshape(-1, 1).long(), 1.0)
                dtyp = _gemm.gemm_tn(oh, g2.contiguous(), out_dtype=torch.float32)[:T].to(dt)
            else:
                dtyp = _table_grad(g2, tt, T).to(dt)
        return dtab, None, dpos, dtyp, None


def embedding(table: torch.Tensor, ids: torch.Tensor, pos: torch.Tensor | None = None,
              typ: torch.Tensor | None = None, token_type_ids: torch.Tensor | None = None) -> torch.Tensor:
    """``table[ids] (+ pos[arange(S)] + typ[token_type_ids])`` -> ``[..., D]``; position ``j`` of the
    last axis of ``ids`` takes ``pos[j]``."""
    if typ is not None and token_type_ids is None:
        token_type_ids = torch.zeros_like(ids)
    return _EmbFn.apply(table, ids, pos, typ, token_type_ids)


def unique_segments(ids: torch.Tensor):
    """Sorts ids and returns (unique ids, permutation, segment offsets [n_unique + 1])."""
    flat = ids.reshape(-1).long()
    sorted_ids, perm = torch.sort(flat, stable=True)
    uniq, counts = torch.unique_consecutive(sorted_ids, return_counts=True)
    seg = torch.zeros(uniq.numel() + 1, dtype=torch.long, device=ids.device)
    seg[1:] = torch.cumsum(counts, 0)
    return uniq, perm, seg


def segment_sum(g: torch.Tensor, perm: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
    """fp32 ``[n_unique, D]``: sum of the rows of ``g`` that share an id, in sorted order."""
    n_u = seg.numel() - 1
    D = g.shape[1]
    out = torch.empty(n_u, D, dtype=torch.float32, device=g.device)
    if g.is_cuda:
        g = g.contiguous()
        rc = _lib.kernels().dtf_emb_segment_sum(_lib.ptr(g), _lib.ptr(perm), _lib.ptr(seg), _lib.ptr(out),
                                                ctypes.c_longlong(n_u), D, _lib.dtype_code(g.dtype),
                                                _lib.stream_ptr(g.device))
        _lib.check(rc, "emb_segment_sum")
        return out
    gs = g.float()[perm]
    out.zero_()
    which = torch.repeat_interleave(torch.arange(n_u, device=g.device), seg[1:] - seg[:-1])
    out.index_add_(0, which, gs)
    return out


def sparse_apply(table, m, v, grad, uids, hyper, kind: int, row_base: int = 0) -> None:
    """In-place optimizer update of rows ``uids - row_base`` of an fp32 table; ``hyper`` is the
    8-float tensor {lr, bc1, bc2, eps, beta1, beta2, grad_scale, 0}."""
    n_u, D = grad.shape
    if table.is_cuda:
        rc = _lib.kernels().dtf_emb_sparse_apply(_lib.ptr(table), _lib.ptr(m), _lib.ptr(v), _lib.ptr(grad),
                                                 _lib.ptr(uids), _lib.ptr(hyper), ctypes.c_longlong(n_u), D,
                                                 ctypes.c_longlong(row_base), ctypes.c_longlong(table.shape[0]), kind,
                                                 _lib.stream_ptr(table.device))
        _lib.check(rc, "emb_sparse_apply")
        return
    lr, bc1, bc2, eps, b1, b2, gs = (float(x) for x in hyper[:7])
    loc = uids - row_base
    ok = (loc >= 0) & (loc < table.shape[0])
    loc, g = loc[ok], grad[ok] * gs
    if kind == SGD:
        table[loc] -= lr * g
    elif kind == ADAGRAD:
        v[loc] += g * g
        table[loc] -= lr * g / (v[loc].sqrt() + eps)
    else:
        m[loc] = b1 * m[loc] + (1 - b1) * g
        v[loc] = b2 * v[loc] + (1 - b2) * g * g
        table[loc] -= lr * (bc2 ** 0.5) / bc1 * m[loc] / (v[loc].sqrt() + eps)
