"""Device-resident Philox state for the fused dropout kernels.

A dropout call draws its mask from ``philox4x32_10(seed, offset + element_index / 4)``. The
64-bit ``offset`` lives on the device and is advanced by a one-thread kernel after every call,
so (a) a call never needs a host round trip, (b) a step captured into a hipGraph draws a fresh
mask on every replay, and (c) the backward pass regenerates the forward mask from a snapshot of
the offset instead of reading a stored mask.

``philox_reference`` is a NumPy implementation of the same generator for the tests.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

_states: dict = {}


class PhiloxState:
    def __init__(self, device, seed: int | None = None):
        self.device = torch.device(device)
        self.seed = int(torch.initial_seed() if seed is None else seed) & ((1 << 64) - 1)
        self.offset = torch.zeros(1, dtype=torch.int64, device=self.device)

    def snapshot(self) -> torch.Tensor:
        """A copy of the current offset (device tensor) for the backward pass of this call."""
        return self.offset.clone()

    def advance(self, n_elements: int) -> None:
        inc = (int(n_elements) + 3) // 4
        rc = _lib.kernels().dtf_rng_tick(_lib.ptr(self.offset), ctypes.c_ulonglong(inc), _lib.stream_ptr(self.device))
        _lib.check(rc, "rng_tick")


def state_for(device) -> PhiloxState:
    device = torch.device(device)
    key = (device.type, device.index)
    st = _states.get(key)
    if st is None or st.seed != (int(torch.initial_seed()) & ((1 << 64) - 1)):
        st = _states[key] = PhiloxState(device)
    return st


def philox_reference(seed: int, counters: np.ndarray) -> np.ndarray:
    """``[len(counters), 4]`` uint32 outputs of Philox4x32-10 with key ``seed`` and counter
    ``(ctr_lo, ctr_hi, 0, 0)``."""
    M32 = np.uint64(0xFFFFFFFF)
    ctr = counters.astype(np.uint64)
    c0, c1 = ctr & M32, ctr >> np.uint64(32)
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    A, B = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    for _ in range(10):
        p0, p1 = A * c0, B * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & M32
        hi1, lo1 = p1 >> np.uint64(32), p1 & M32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & M32, lo1, (hi0 ^ c3 ^ k1) & M32, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.uint32)


def dropout_keep_mask_reference(seed: int, offset: int, numel: int, p: float) -> np.ndarray:
    """Boolean keep mask of ``numel`` elements (linear order) as the kernels compute it."""
    groups = (numel + 3) // 4
    r = philox_reference(seed, np.arange(groups, dtype=np.uint64) + np.uint64(offset)).reshape(-1)[:numel]
    thresh = min(int(p * 4294967296.0), 4294967295) if p > 0 else 0
    return r >= np.uint32(thresh)
