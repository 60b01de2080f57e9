"""Whole-step hipGraph capture.

A training step of a small model is launch-bound: the reference MLP at batch 100 is ~60
kernels of a few microseconds each, and the Python + HIP launch path costs more than the GPU
work. Every kernel of this framework takes raw pointers and a stream and never allocates or
synchronises, the optimizer's step counter and bias corrections live on the device, and the
gradient arena has fixed addresses, so an entire ``train_step`` -- forward, backward, bucket
packing, collectives, optimizer -- can be captured once and replayed with one launch.

``GraphedStep`` runs ``warmup`` eager steps (library autotuning, workspace allocation), then
captures one graph per input signature (shapes + dtypes) and replays it; an unseen signature
(e.g. a short last batch) falls back to a fresh capture of its own. The returned loss is a
static tensor that the next replay overwrites.

The reference has no counterpart: it runs TensorFlow eagerly, one op at a time
(reference centralized/client.py:16-19,83).
"""
from __future__ import annotations

import torch


def _sig(t):
    if isinstance(t, (tuple, list)):
        return tuple(_sig(u) for u in t)
    return (tuple(t.shape), t.dtype)


def _clone_static(t):
    if isinstance(t, (tuple, list)):
        return tuple(_clone_static(u) for u in t)
    return t.detach().clone()


def _copy_into(dst, src) -> None:
    if isinstance(dst, tuple):
        for d, s in zip(dst, src):
            _copy_into(d, s)
    else:
        dst.copy_(src, non_blocking=True)


class GraphedStep:
    def __init__(self, step_fn, device, warmup: int = 3, max_graphs: int = 4):
        self.step_fn = step_fn
        self.device = torch.device(device)
        self.warmup = warmup
        self.max_graphs = max_graphs
        self._eager_steps = 0
        self._graphs: dict = {}
        self._pool = None
        self.replays = 0
        self.captures = 0

    @property
    def active(self) -> bool:
        return self.device.type == "cuda"

    def __call__(self, x, y):
        if not self.active:
            return self.step_fn(x, y)
        if self._eager_steps < self.warmup:
            # eager steps on a side stream, as capture requires of any state they allocate
            self._eager_steps += 1
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                loss = self.step_fn(x, y)
            cur.wait_stream(side)
            return loss
        key = (_sig(x), _sig(y))
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                return self.step_fn(x, y)
            entry = self._capture(x, y)
            self._graphs[key] = entry
            return entry[3]          # the capture pass does not execute: replay once below
        g, sx, sy, loss = entry
        _copy_into(sx, x)
        _copy_into(sy, y)
        g.replay()
        self.replays += 1
        return loss

    def _capture(self, x, y):
        sx, sy = _clone_static(x), _clone_static(y)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._pool):
            loss = self.step_fn(sx, sy)
        if self._pool is None:
            self._pool = g.pool()
        self.captures += 1
        g.replay()                   # capture records only; run the step it stands for
        self.replays += 1
        return g, sx, sy, loss
