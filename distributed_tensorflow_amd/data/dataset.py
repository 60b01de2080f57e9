"""A small tensor-slice dataset pipeline: ``from_tensor_slices -> shard -> batch -> shuffle``.

This is the subset of ``tf.data`` the reference's example ``dataset_fn`` uses
(reference initializer.py:43-48): ``shard(n, i)`` keeps every n-th element,
``batch(b)`` groups consecutive elements, ``shuffle(buf)`` permutes whatever
the current elements are -- so calling it after ``batch`` shuffles *batches*,
exactly as in the reference.

The pipeline is evaluated lazily as index arithmetic over the source arrays,
which stay where they are (host memory, pinned memory or GPU memory). A batch
is materialised with ONE gather; ``to(device)`` makes the whole dataset GPU
resident so an epoch issues no host-to-device copies at all, and ``prefetch``
overlaps the gather/copy of the next batch on a side stream.
"""
from __future__ import annotations

from typing import Iterator, Sequence

import numpy as np
import torch


class Dataset:
    def __init__(self, tensors: Sequence[torch.Tensor], index: torch.Tensor | None = None, batch_size: int | None = None,
                 drop_remainder: bool = False, shuffle_buffer: int | None = None, shuffle_level: str | None = None,
                 seed: int = 0, repeat: int | None = 1, prefetch: int = 0):
        self.tensors = tuple(tensors)
        n = self.tensors[0].shape[0]
        for t in self.tensors:
            if t.shape[0] != n:
                raise ValueError("all tensors must share the first dimension")
        self.index = index if index is not None else torch.arange(n)
        self.batch_size = batch_size
        self.drop_remainder = drop_remainder
        self.shuffle_buffer = shuffle_buffer
        self.shuffle_level = shuffle_level  # 'element' | 'batch' | None
        self.seed = seed
        self.repeat_count = repeat
        self.prefetch_depth = prefetch
        self._epoch = 0

    # ------------------------------------------------------------ constructors
    @classmethod
    def from_tensor_slices(cls, tensors) -> "Dataset":
        if isinstance(tensors, (torch.Tensor, np.ndarray)):
            tensors = (tensors,)
        ts = [torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t for t in tensors]
        return cls(ts)

    def _clone(self, **kw) -> "Dataset":
        d = Dataset.__new__(Dataset)
        d.__dict__.update(self.__dict__)
        d.__dict__.update(kw)
        return d

    # ------------------------------------------------------------ transformations
    def shard(self, num_shards: int, index: int) -> "Dataset":
        if self.batch_size is not None:
            raise ValueError("shard() must come before batch()")
        if not 0 <= index < num_shards:
            raise ValueError("shard index out of range")
        return self._clone(index=self.index[index::num_shards])

    def take(self, n: int) -> "Dataset":
        if self.batch_size is not None:
            return self._clone(index=self.index[: n * self.batch_size])
        return self._clone(index=self.index[:n])

    def batch(self, batch_size: int, drop_remainder: bool = False) -> "Dataset":
        return self._clone(batch_size=int(batch_size), drop_remainder=drop_remainder)

    def shuffle(self, buffer_size: int, seed: int | None = None) -> "Dataset":
        level = "batch" if self.batch_size is not None else "element"
        return self._clone(shuffle_buffer=int(buffer_size), shuffle_level=level,
                           seed=self.seed if seed is None else seed)

    def repeat(self, count: int | None = None) -> "Dataset":
        return self._clone(repeat_count=count)

    def prefetch(self, depth: int = 2) -> "Dataset":
        return self._clone(prefetch_depth=int(depth))

    def to(self, device) -> "Dataset":
        """Moves the SOURCE tensors to ``device`` (dataset becomes device resident)."""
        return self._clone(tensors=tuple(t.to(device, non_blocking=True) for t in self.tensors))

    def pin(self) -> "Dataset":
        return self._clone(tensors=tuple(t.pin_memory() if not t.is_cuda else t for t in self.tensors))

    # ------------------------------------------------------------ introspection
    def num_elements(self) -> int:
        return int(self.index.numel())

    def num_batches(self) -> int:
        n = self.num_elements()
        if self.batch_size is None:
            return n
        return n // self.batch_size if self.drop_remainder else -(-n // self.batch_size)

    def __len__(self) -> int:
        return self.num_batches() * (self.repeat_count or 1)

    # ------------------------------------------------------------ iteration
    @staticmethod
    def _buffer_shuffle(n: int, buf: int, rng: np.random.Generator) -> np.ndarray:
        """Order in which a tf.data-style shuffle buffer of size ``buf`` emits ``n`` items."""
        if buf >= n:
            return rng.permutation(n)
        out = np.empty(n, dtype=np.int64)
        pool = list(range(buf))
        nxt = buf
        for i in range(n):
            j = int(rng.integers(len(pool)))
            out[i] = pool[j]
            if nxt < n:
                pool[j] = nxt
                nxt += 1
            else:
                pool[j] = pool[-1]
                pool.pop()
        return out

    def _epoch_batches(self, epoch: int) -> list[torch.Tensor]:
        idx = self.index
        rng = np.random.default_rng((self.seed, epoch))
        if self.shuffle_level == "element":
            idx = idx[torch.from_numpy(self._buffer_shuffle(idx.numel(), self.shuffle_buffer, rng))]
        if self.batch_size is None:
            return [idx[i:i + 1] for i in range(idx.numel())]
        nb = self.num_batches()
        batches = [idx[i * self.batch_size:(i + 1) * self.batch_size] for i in range(nb)]
        if self.shuffle_level == "batch":
            order = self._buffer_shuffle(nb, self.shuffle_buffer, rng)
            batches = [batches[i] for i in order]
        return batches

    def _gather(self, sel: torch.Tensor, device) -> tuple[torch.Tensor, ...]:
        out = []
        for t in self.tensors:
            s = sel.to(t.device) if t.is_cuda else sel
            b = t.index_select(0, s)
            if self.batch_size is None:
                b = b[0]
            if device is not None and b.device != device:
                b = b.to(device, non_blocking=True)
            out.append(b)
        return tuple(out)

    def iterate(self, device=None) -> Iterator[tuple[torch.Tensor, ...]]:
        device = torch.device(device) if device is not None else None
        reps = self.repeat_count
        ep = 0
        while reps is None or ep < reps:
            batches = self._epoch_batches(self._epoch)
            self._epoch += 1
            ep += 1
            if self.prefetch_depth > 0 and device is not None and device.type == "cuda":
                yield from self._prefetched(batches, device)
            else:
                for sel in batches:
                    yield self._gather(sel, device)

    def _prefetched(self, batches, device):
        side = torch.cuda.Stream(device)
        queue: list = []
        it = iter(batches)

        def launch():
            sel = next(it, None)
            if sel is None:
                return False
            with torch.cuda.stream(side):
                item = self._gather(sel, device)
                ev = torch.cuda.Event()
                ev.record(side)
            queue.append((item, ev))
            return True

        for _ in range(self.prefetch_depth):
            if not launch():
                break
        while queue:
            item, ev = queue.pop(0)
            torch.cuda.current_stream(device).wait_event(ev)
            for t in item:
                t.record_stream(torch.cuda.current_stream(device))
            launch()
            yield item

    def __iter__(self):
        return self.iterate(None)
