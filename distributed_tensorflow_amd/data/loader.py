"""Python face of the native batch loader (``csrc/runtime/loader.cpp``): worker threads gather rows
by index into (pinned) staging buffers and convert uint8 pixels to float32 in [0, 1]."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from ..ops import _lib


def _rt():
    lib = _lib.runtime()
    if not getattr(lib, "_loader_declared", False):
        lib.dtf_loader_create.restype = ctypes.c_void_p
        lib.dtf_loader_create.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int]
        lib.dtf_loader_submit.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        lib.dtf_loader_wait.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]
        lib.dtf_loader_release.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.dtf_loader_destroy.argtypes = [ctypes.c_void_p]
        lib.dtf_loader_is_pinned.argtypes = [ctypes.c_void_p]
        lib._loader_declared = True
    return lib


class NativeLoader:
    def __init__(self, x: np.ndarray, y: np.ndarray | None, max_batch: int, to_float: bool = True, slots: int = 4,
                 threads: int = 2, pinned: bool = True):
        self.x = np.ascontiguousarray(x)
        self.y = np.ascontiguousarray(y) if y is not None else None
        if to_float and self.x.dtype != np.uint8:
            raise TypeError("to_float converts uint8 pixels; pass uint8 data or to_float=False")
        self.row_shape = self.x.shape[1:]
        self.x_row = int(np.prod(self.row_shape)) * self.x.itemsize
        self.y_row = self.y.itemsize * int(np.prod(self.y.shape[1:])) if self.y is not None else 0
        self.to_float = to_float
        self.h = _rt().dtf_loader_create(self.x.ctypes.data, self.x.shape[0], self.x_row,
                                         self.y.ctypes.data if self.y is not None else None, self.y_row, slots,
                                         max_batch, int(to_float), threads, int(pinned))
        if not self.h:
            raise RuntimeError("native loader could not be created")
        self.pinned = bool(_rt().dtf_loader_is_pinned(self.h))

    def submit(self, indices) -> int:
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.int64))
        s = _rt().dtf_loader_submit(self.h, idx.ctypes.data, idx.size)
        if s < 0:
            raise RuntimeError("no free loader slot (release one first)" if s == -2 else "bad batch")
        return s

    def take(self, slot: int):
        """Blocks until the batch is assembled; returns tensors that alias the staging buffers
        (valid until :meth:`release`)."""
        xp, yp, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        if _rt().dtf_loader_wait(self.h, slot, ctypes.byref(xp), ctypes.byref(yp), ctypes.byref(n)) != 0:
            raise RuntimeError("bad loader slot")
        n = n.value
        xdt = np.float32 if self.to_float else self.x.dtype
        nx = n * int(np.prod(self.row_shape))
        xb = np.ctypeslib.as_array(ctypes.cast(xp, ctypes.POINTER(np.ctypeslib.as_ctypes_type(xdt))), shape=(nx,))
        bx = torch.from_numpy(xb).view(n, *self.row_shape)
        by = None
        if self.y is not None:
            ny = n * int(np.prod(self.y.shape[1:]))
            yb = np.ctypeslib.as_array(ctypes.cast(yp, ctypes.POINTER(np.ctypeslib.as_ctypes_type(self.y.dtype))),
                                       shape=(ny,))
            by = torch.from_numpy(yb).view(n, *self.y.shape[1:])
        return bx, by

    def release(self, slot: int) -> None:
        _rt().dtf_loader_release(self.h, slot)

    def close(self) -> None:
        if self.h:
            _rt().dtf_loader_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
