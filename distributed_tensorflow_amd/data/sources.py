"""Dataset sources: MNIST / CIFAR-10 from local files when present, otherwise
deterministic synthetic stand-ins of the same shape, dtype and class structure.

The reference downloads MNIST through ``tf.keras.datasets.mnist.load_data()``
(reference initializer.py:26,32). There is no network on the training nodes
this framework targets, so :func:`load_mnist` looks for ``mnist.npz`` (the
Keras cache file: arrays ``x_train, y_train, x_test, y_test``) under
``$DTF_DATA_DIR``, ``~/.keras/datasets`` and ``./datasets``; failing that it
synthesises a 10-class problem with the MNIST shapes (60000/10000 x 28 x 28
uint8) whose classes are separable but noisy, so accuracy comparisons between
strategies remain meaningful. Every call with the same seed returns the same
arrays, in every process.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch


def _search(name: str) -> Path | None:
    roots = [os.environ.get("DTF_DATA_DIR"), str(Path.home() / ".keras" / "datasets"), "./datasets", "."]
    for r in roots:
        if r and (Path(r) / name).exists():
            return Path(r) / name
    return None


def _synthetic_classification(n_train: int, n_test: int, shape: tuple[int, ...], classes: int, seed: int,
                              noise: float = 0.35):
    """Class prototypes (smooth random images) + per-sample noise and random shifts -> uint8 images."""
    rng = np.random.default_rng(seed)
    h, w = shape[0], shape[1]
    ch = shape[2] if len(shape) == 3 else 1
    protos = rng.normal(size=(classes, h // 4 + 1, w // 4 + 1, ch)).astype(np.float32)
    protos = np.repeat(np.repeat(protos, 4, axis=1), 4, axis=2)[:, :h, :w, :]
    protos = (protos - protos.min()) / (protos.max() - protos.min())

    def make(n, rng):
        y = rng.integers(0, classes, size=n).astype(np.uint8)
        x = protos[y]
        sx, sy = rng.integers(-2, 3, size=n), rng.integers(-2, 3, size=n)
        out = np.empty_like(x)
        for s in range(-2, 3):
            for t in range(-2, 3):
                sel = (sx == s) & (sy == t)
                if sel.any():
                    out[sel] = np.roll(np.roll(x[sel], s, axis=1), t, axis=2)
        out = out + noise * rng.normal(size=out.shape).astype(np.float32)
        out = (np.clip(out, 0, 1) * 255).astype(np.uint8)
        if len(shape) == 2:
            out = out[..., 0]
        return out, y

    xtr, ytr = make(n_train, np.random.default_rng(seed + 1))
    xte, yte = make(n_test, np.random.default_rng(seed + 2))
    return (xtr, ytr), (xte, yte)


def _teacher_classification(n_train: int, n_test: int, seed: int, tau: float = 0.03, flip: float = 0.02):
    """A 10-class problem with MNIST's shapes that a strategy can do visibly WORSE on: images are random 7 x 7 fields
    shown at 28 x 28, labels come from a fixed random linear teacher on the 49 field values with Gumbel noise on its
    scores (examples near a decision boundary are mislabelled in proportion to how near) and ``flip`` of all labels
    drawn at random -- in the training AND the test split. No classifier exceeds the teacher's own accuracy on the
    noisy labels (about 0.95); the reference MLP under Adam(1e-3) gets to 0.85-0.88 in three epochs and a learning
    rate ten times too high, or lost updates, cost several points -- unlike the prototype task, which every
    strategy solves to 99.7 %."""
    w = np.random.default_rng(seed).normal(size=(49, 10)).astype(np.float32)

    def make(n, sd):
        r = np.random.default_rng(sd)
        lo = r.integers(0, 256, size=(n, 7, 7), dtype=np.uint8)
        x = np.repeat(np.repeat(lo, 4, axis=1), 4, axis=2)
        scores = (lo.reshape(n, -1).astype(np.float32) / 255 - 0.5) @ w
        y = (scores + tau * r.gumbel(size=scores.shape)).argmax(1)
        y = np.where(r.random(n) < flip, r.integers(0, 10, size=n), y)
        return x, y.astype(np.uint8)

    return make(n_train, seed + 1), make(n_test, seed + 2)


_cache: dict = {}


def load_mnist(seed: int = 1234, synthetic_sizes: tuple[int, int] = (60000, 10000)):
    """((x_train, y_train), (x_test, y_test)) as uint8 arrays, images ``[N, 28, 28]``; third item of
    the return value says where the data came from (``'file:<path>'`` or ``'synthetic'``)."""
    kind = os.environ.get("DTF_SYNTH_MNIST_KIND", "prototypes")     # 'teacher': the noisy-teacher task (convergence studies)
    key = ("mnist", seed, synthetic_sizes, kind)
    if key in _cache:
        return _cache[key]
    p = _search("mnist.npz") if kind != "teacher" else None
    if p is not None:
        with np.load(p, allow_pickle=False) as z:
            out = ((z["x_train"], z["y_train"]), (z["x_test"], z["y_test"]), f"file:{p}")
    elif kind == "teacher":
        tr, te = _teacher_classification(*synthetic_sizes, seed)
        out = (tr, te, "synthetic-teacher")
    else:
        tr, te = _synthetic_classification(*synthetic_sizes, (28, 28), 10, seed)
        out = (tr, te, "synthetic")
    _cache[key] = out
    return out


def load_cifar10(seed: int = 4321, synthetic_sizes: tuple[int, int] = (50000, 10000)):
    """CIFAR-10 shaped data, images ``[N, 32, 32, 3]`` uint8; file name ``cifar10.npz`` with the
    same four array names as the MNIST cache file."""
    key = ("cifar10", seed, synthetic_sizes)
    if key in _cache:
        return _cache[key]
    p = _search("cifar10.npz")
    if p is not None:
        with np.load(p, allow_pickle=False) as z:
            out = ((z["x_train"], z["y_train"].reshape(-1)), (z["x_test"], z["y_test"].reshape(-1)), f"file:{p}")
    else:
        tr, te = _synthetic_classification(*synthetic_sizes, (32, 32, 3), 10, seed, noise=0.5)
        out = (tr, te, "synthetic")
    _cache[key] = out
    return out


def synthetic_images(batch: int, height: int, width: int, channels: int, classes: int, device, dtype=torch.bfloat16,
                     seed: int = 0, channels_last: bool = True):
    """One device-resident random batch ``(x NHWC, y)`` of the given shape (throughput benchmarks)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch, height, width, channels, generator=g, dtype=torch.float32).to(device=device, dtype=dtype)
    y = torch.randint(0, classes, (batch,), generator=g, dtype=torch.int64).to(device)
    return x, y


def synthetic_tokens(batch: int, seq_len: int, vocab: int, device, mask_prob: float = 0.15, seed: int = 0):
    """Random MLM batch: ``(input_ids, labels)`` with ``labels = -100`` at unmasked positions."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    ids = torch.randint(0, vocab, (batch, seq_len), generator=g, dtype=torch.int64)
    mask = torch.rand(batch, seq_len, generator=g) < mask_prob
    labels = torch.where(mask, ids, torch.full_like(ids, -100))
    ids = torch.where(mask, torch.full_like(ids, 103 % vocab), ids)
    return ids.to(device), labels.to(device)
