"""ctypes loader for the in-tree native libraries.

The HIP kernels are exported through a C ABI (raw device pointers plus a
``hipStream_t``), so the Python side needs no compiled binding layer: a call is
``lib.fn(tensor.data_ptr(), ..., stream)``. ctypes releases the GIL for the
duration of the call, which the parameter-server threads rely on.

Tensors on a GPU never fall back to an eager implementation: if the kernel
library is missing, :func:`kernels` raises.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

_LIB_DIR = Path(__file__).resolve().parent.parent / "lib"
_kernels = None
_runtime = None

F32, BF16 = 0, 1
_DTYPE_CODE = {torch.float32: F32, torch.bfloat16: BF16}


class NativeLibraryMissing(RuntimeError):
    pass


def _load(name: str) -> ctypes.CDLL:
    path = _LIB_DIR / name
    if not path.exists():
        if os.environ.get("DTF_AUTO_BUILD", "1") == "1":
            try:
                from ..csrc import build

                (build.build_kernels if "kernels" in name else build.build_runtime)()
            except Exception as e:  # pragma: no cover - depends on toolchain
                raise NativeLibraryMissing(f"{path} is missing and could not be built: {e}") from e
        if not path.exists():
            raise NativeLibraryMissing(
                f"{path} is missing; run `python -m distributed_tensorflow_amd.csrc.build`")
    return ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)


def kernels() -> ctypes.CDLL:
    """The HIP kernel library (``libdtf_kernels.so``)."""
    global _kernels
    if _kernels is None:
        _kernels = _load("libdtf_kernels.so")
        cap = os.environ.get("DTF_STREAM_BLOCKS")       # A/B switch: grid cap of the streaming kernels
        if cap:
            _kernels.dtf_set_stream_blocks(int(cap))
        if os.environ.get("DTF_LN_BWD_BLOCKS"):         # A/B switch: block cap of the LayerNorm backward pass
            _kernels.dtf_ln_set_bwd_blocks(int(os.environ["DTF_LN_BWD_BLOCKS"]))
        if os.environ.get("DTF_GEMM_STAGES"):           # A/B switch: main loop of the 128x128 GEMM (1, 2: LDS buffers of the
            _kernels.dtf_gemm_set_stages(int(os.environ["DTF_GEMM_STAGES"]))   # register-staged loop; 3: LDS-DMA)
        if os.environ.get("DTF_GEMM_DMA_MIN_TILES"):    # k-tiles per block from which the LDS-DMA loop is chosen
            _kernels.dtf_gemm_set_dma_min_tiles(int(os.environ["DTF_GEMM_DMA_MIN_TILES"]))
        if os.environ.get("DTF_GEMM_XCD_SPLITS") == "0":  # A/B switch: split-K tiles dealt round-robin over the XCDs
            _kernels.dtf_gemm_set_xcd_splits(0)
        if os.environ.get("DTF_GEMM_LEAN"):             # A/B switch: 0 = general store phase everywhere,
            _kernels.dtf_gemm_set_lean(int(os.environ["DTF_GEMM_LEAN"]))   # 2 = lean for plain products only
        rows = os.environ.get("DTF_BN_APPLY")           # A/B switch "rows,nt": items per thread of the batch-norm
        if rows:                                        # apply kernels, non-temporal loads on / off
            r, nt = rows.split(",")
            _kernels.dtf_set_bn_apply_rows(int(r), int(nt))
    return _kernels


def runtime() -> ctypes.CDLL:
    """The host runtime library (``libdtf_runtime.so``)."""
    global _runtime
    if _runtime is None:
        _runtime = _load("libdtf_runtime.so")
    return _runtime


def dtype_code(t: torch.dtype) -> int:
    try:
        return _DTYPE_CODE[t]
    except KeyError:
        raise TypeError(f"unsupported dtype {t}; the kernels take float32 and bfloat16") from None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None) -> ctypes.c_void_p:
    """The current HIP stream of ``device`` as a raw handle. Called once per kernel launch: the raw getter costs a
    fraction of a microsecond where ``torch.cuda.current_stream`` builds a Python object every time (5 us, a quarter
    of a millisecond per ResNet-20 step)."""
    if _raw_stream is not None:
        if device is None:
            idx = torch.cuda.current_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = device.index if device.index is not None else torch.cuda.current_device()
        return ctypes.c_void_p(_raw_stream(idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t: torch.Tensor | None) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed with HIP/launch status {code}")
