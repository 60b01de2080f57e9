"""Keras-compatible weight import / export.

The interchange format is the one the reference uses on the wire and in
memory: the ordered list of numpy arrays returned by Keras'
``model.get_weights()`` (reference centralized/server.py:96,150;
client.py:72,90). Per layer, in model order: trainable weights then
non-trainable ones, in Keras layouts:

===================  ==========================================  =======================
layer                Keras arrays                                 torch storage
===================  ==========================================  =======================
Dense / Linear       kernel ``(in, out)``, bias ``(out,)``         weight ``(out, in)``
Conv2D / Conv2d      kernel ``(H, W, I, O)``, bias ``(O,)``        weight ``(O, I, H, W)``
BatchNormalization   gamma, beta, moving_mean, moving_variance    weight, bias, running_*
LayerNormalization   gamma, beta                                  weight, bias
Embedding            embeddings ``(vocab, dim)``                  weight ``(vocab, dim)``
===================  ==========================================  =======================

Modules that implement ``keras_weights()`` / ``set_keras_weights()`` speak for
their whole subtree; other leaf modules are converted by the table above;
anything else contributes its own parameters then buffers unchanged.

On disk the list is an ``.npz`` (``weight_000``, ``weight_001``, ... plus a
``names`` array); when ``h5py`` is importable, ``.h5`` files use the Keras
``model_weights`` group layout so that ``keras.Model.load_weights`` can read
them.
"""
from __future__ import annotations

import os
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def _assign(p: torch.Tensor, a: np.ndarray, what: str) -> None:
    a = np.ascontiguousarray(a)
    if tuple(a.shape) != tuple(p.shape):
        raise ValueError(f"{what}: expected shape {tuple(p.shape)}, got {tuple(a.shape)}")
    with torch.no_grad():
        src = torch.from_numpy(a)
        p.copy_(src.to(p.dtype))
        # a bf16 parameter that lives in a strategy's flat arenas is a view of the bf16 shadow; the optimizer's
        # source of truth is its fp32 master slice (engine/flat.py), which takes the unrounded value
        master = getattr(p, "_dtf_master", None)
        if master is not None:
            master.copy_(src.to(master.dtype))


def _leaf_export(m: nn.Module) -> list[np.ndarray] | None:
    if isinstance(m, nn.Linear):
        out = [_np(m.weight).T.copy()]
        if m.bias is not None:
            out.append(_np(m.bias))
        return out
    if isinstance(m, nn.Conv2d):
        out = [np.ascontiguousarray(_np(m.weight).transpose(2, 3, 1, 0))]
        if m.bias is not None:
            out.append(_np(m.bias))
        return out
    if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
        out = []
        if m.affine:
            out += [_np(m.weight), _np(m.bias)]
        if m.track_running_stats:
            out += [_np(m.running_mean), _np(m.running_var)]
        return out
    if isinstance(m, nn.LayerNorm):
        return [_np(m.weight), _np(m.bias)] if m.elementwise_affine else []
    if isinstance(m, nn.Embedding):
        return [_np(m.weight)]
    return None


def _leaf_import(m: nn.Module, ws: Sequence[np.ndarray], name: str) -> int | None:
    if isinstance(m, nn.Linear):
        _assign(m.weight, np.asarray(ws[0]).T, name + ".kernel")
        if m.bias is not None:
            _assign(m.bias, np.asarray(ws[1]), name + ".bias")
            return 2
        return 1
    if isinstance(m, nn.Conv2d):
        _assign(m.weight, np.asarray(ws[0]).transpose(3, 2, 0, 1), name + ".kernel")
        if m.bias is not None:
            _assign(m.bias, np.asarray(ws[1]), name + ".bias")
            return 2
        return 1
    if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
        k = 0
        if m.affine:
            _assign(m.weight, np.asarray(ws[0]), name + ".gamma")
            _assign(m.bias, np.asarray(ws[1]), name + ".beta")
            k = 2
        if m.track_running_stats:
            _assign(m.running_mean, np.asarray(ws[k]), name + ".moving_mean")
            _assign(m.running_var, np.asarray(ws[k + 1]), name + ".moving_variance")
            k += 2
        return k
    if isinstance(m, nn.LayerNorm):
        if m.elementwise_affine:
            _assign(m.weight, np.asarray(ws[0]), name + ".gamma")
            _assign(m.bias, np.asarray(ws[1]), name + ".beta")
            return 2
        return 0
    if isinstance(m, nn.Embedding):
        _assign(m.weight, np.asarray(ws[0]), name + ".embeddings")
        return 1
    return None


def _walk(module: nn.Module, prefix: str = ""):
    """Yields (name, module, kind) for every weight-owning unit in registration order."""
    if hasattr(module, "keras_weights") and hasattr(module, "set_keras_weights") and prefix:
        yield prefix, module, "protocol"
        return
    if _leaf_export(module) is not None:
        yield prefix, module, "leaf"
        return
    own = list(module._parameters.items()) + list(module._buffers.items())
    if any(v is not None for _, v in own):
        yield prefix, module, "raw"
    for n, child in module.named_children():
        yield from _walk(child, f"{prefix}.{n}" if prefix else n)


def _raw_tensors(m: nn.Module):
    out = [(n, p) for n, p in m._parameters.items() if p is not None]
    out += [(n, b) for n, b in m._buffers.items() if b is not None and b.is_floating_point()]
    return out


def named_weights(module: nn.Module) -> list[tuple[str, np.ndarray]]:
    out: list[tuple[str, np.ndarray]] = []
    for name, m, kind in _walk(module):
        if kind == "protocol":
            arrs = m.keras_weights()
        elif kind == "leaf":
            arrs = _leaf_export(m)
        else:
            arrs = [_np(t) for _, t in _raw_tensors(m)]
        out += [(f"{name}/{i}", a) for i, a in enumerate(arrs)]
    return out


def get_weights(module: nn.Module) -> list[np.ndarray]:
    return [a for _, a in named_weights(module)]


def set_weights(module: nn.Module, weights: Sequence[np.ndarray]) -> None:
    ws = list(weights)
    expected = len(named_weights(module))
    if len(ws) != expected:
        raise ValueError(f"set_weights: model takes {expected} arrays, {len(ws)} were given")
    i = 0
    for name, m, kind in _walk(module):
        if kind == "protocol":
            k = m.set_keras_weights(ws[i:])
        elif kind == "leaf":
            k = _leaf_import(m, ws[i:], name)
        else:
            k = 0
            for n, t in _raw_tensors(m):
                _assign(t, np.asarray(ws[i + k]), f"{name}.{n}")
                k += 1
        i += k
    if i != len(ws):
        raise ValueError(f"set_weights: model takes {i} arrays, {len(ws)} were given")
    sync_masters(module)


def sync_masters(module: nn.Module) -> None:
    """After a write to the parameters of a model whose bf16 parameters are views of a strategy's shadow arena:
    bring the fp32 master slices (the optimizer's source of truth) in line, so that the next optimizer step does
    not undo the write. Slices that :func:`_assign` already wrote at full precision are left alone."""
    with torch.no_grad():
        for p in module.parameters():
            master = getattr(p, "_dtf_master", None)
            if master is not None and not torch.equal(master.to(p.dtype), p.data):
                master.copy_(p.data.to(master.dtype))


def save_weights(module: nn.Module, path: str) -> None:
    nw = named_weights(module)
    if path.endswith((".h5", ".hdf5")):
        try:
            import h5py  # type: ignore
        except ImportError as e:
            raise RuntimeError("writing Keras HDF5 needs h5py, which is not installed; use an .npz path") from e
        with h5py.File(path, "w") as f:
            g = f.create_group("model_weights")
            layers: dict[str, list[tuple[str, np.ndarray]]] = {}
            for n, a in nw:
                layer, idx = n.rsplit("/", 1)
                layers.setdefault(layer, []).append((f"{layer}/weight_{idx}:0", a))
            g.attrs["layer_names"] = [k.encode() for k in layers]
            for layer, items in layers.items():
                lg = g.create_group(layer)
                lg.attrs["weight_names"] = [k.encode() for k, _ in items]
                for k, a in items:
                    lg.create_dataset(k, data=a)
        return
    arrays = {f"weight_{i:03d}": a for i, (_, a) in enumerate(nw)}
    arrays["names"] = np.array([n for n, _ in nw])
    if not path.endswith(".npz"):
        path += ".npz"
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:       # written beside the target and renamed: a crash never leaves a torn file
        np.savez(f, **arrays)
    os.replace(tmp, path)


def load_weight_list(path: str) -> list[np.ndarray]:
    if path.endswith((".h5", ".hdf5")):
        try:
            import h5py  # type: ignore
        except ImportError as e:
            raise RuntimeError("reading Keras HDF5 needs h5py, which is not installed") from e
        out = []
        with h5py.File(path, "r") as f:
            g = f["model_weights"] if "model_weights" in f else f
            for layer in g.attrs["layer_names"]:
                lg = g[layer.decode() if isinstance(layer, bytes) else layer]
                for wn in lg.attrs["weight_names"]:
                    out.append(np.asarray(lg[wn.decode() if isinstance(wn, bytes) else wn]))
        return out
    with np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False) as z:
        keys = sorted((k for k in z.files if k.startswith("weight_")), key=lambda k: int(k.split("_", 1)[1]))
        return [z[k] for k in keys]


def load_weights(module: nn.Module, path: str) -> None:
    set_weights(module, load_weight_list(path))
