"""Checkpoint / resume.

The reference keeps weights only in memory (its nearest thing to a checkpoint is the
``get_weights()`` list it sends over the wire, reference centralized/server.py:96,150). A
checkpoint here is a directory:

``model.pt``        rank 0: the fp32 master arena, the slot table (name -> offset, shape), every
                    floating buffer (batch-norm statistics), the step counter;
``weights.npz``     rank 0: the same weights as the Keras-ordered array list (importable by
                    ``Model.load_weights`` / readable with numpy);
``optim_rank{r}.pt`` every rank: optimizer moments and hyper-parameters for the arena ranges that
                    rank owns (the whole arena for mirrored strategies, its shards for the
                    parameter-server strategies), plus its shard of every row-sharded embedding.

Files are written to a temporary name and renamed, so a crash never leaves a torn checkpoint;
they hold only tensors and plain Python containers and are loaded with ``weights_only=True``.
"""
from __future__ import annotations

import os

import torch

from ..api import keras_weights as kw


def _atomic_save(obj, path: str) -> None:
    tmp = path + ".tmp"
    torch.save(obj, tmp)
    os.replace(tmp, path)


def _rng_state(device) -> dict:
    """Random-number state that a resumed run must continue from for dropout masks to repeat: torch's CPU (and
    device) generator and the device-resident Philox offset of the fused dropout kernels."""
    out = {"cpu": torch.get_rng_state()}
    if device.type == "cuda":
        out["cuda"] = torch.cuda.get_rng_state(device)
        from ..ops import rng as _rng
        st = _rng._states.get((device.type, device.index))
        if st is not None:
            out["philox_seed"] = torch.tensor([st.seed & ((1 << 63) - 1), st.seed >> 63], dtype=torch.int64)
            out["philox_offset"] = st.offset.detach().cpu()
    return out


def _set_rng_state(state: dict, device) -> None:
    torch.set_rng_state(state["cpu"])
    if device.type == "cuda" and "cuda" in state:
        torch.cuda.set_rng_state(state["cuda"], device)
        if "philox_offset" in state:
            from ..ops import rng as _rng
            st = _rng.state_for(device)
            st.offset.copy_(state["philox_offset"].to(device))


def save_checkpoint(directory: str, strategy, step: int | None = None, keras_export: bool = True) -> None:
    os.makedirs(directory, exist_ok=True)
    comm, flat = strategy.comm, strategy.flat
    model = strategy.replica.model
    master = strategy.export_master()           # collective where the master is sharded across the ranks
    if comm.rank == 0:
        slots = {s.name: {"offset": s.offset, "numel": s.numel, "shape": list(s.shape), "stride": list(s.stride)}
                 for s in flat.slots}
        buffers = {n: b.detach().cpu() for n, b in model.named_buffers() if b.is_floating_point() or b.dtype == torch.long}
        _atomic_save({"master": master.detach().cpu(), "slots": slots, "buffers": buffers,
                      "step": int(step if step is not None else strategy.replica.step_count),
                      "strategy": strategy.name, "world": comm.world}, os.path.join(directory, "model.pt"))
        if keras_export:
            kw.save_weights(model, os.path.join(directory, "weights.npz"))
    opts = getattr(strategy, "optimizers", None) or ([strategy.optimizer] if strategy.optimizer is not None else [])
    state = {"optimizers": [o.state_dict() for o in opts],
             "sparse": [m.state_dict_shard() for m in getattr(strategy, "sparse_modules", [])],
             "rng": _rng_state(comm.device)}
    _atomic_save(state, os.path.join(directory, f"optim_rank{comm.rank}.pt"))
    comm.barrier()


def load_checkpoint(directory: str, strategy) -> int:
    """Restores weights, buffers and this rank's optimizer state; returns the saved step."""
    comm, flat = strategy.comm, strategy.flat
    model = strategy.replica.model
    ck = torch.load(os.path.join(directory, "model.pt"), map_location="cpu", weights_only=True)
    master = ck["master"]
    full = torch.zeros(flat.total, dtype=torch.float32)
    for s in flat.slots:  # by name, so a checkpoint survives a different padding / shard count / slot order
        src = ck["slots"][s.name]
        if src["numel"] != s.numel or list(src["shape"]) != list(s.shape):
            raise ValueError(f"checkpoint tensor {s.name} has shape {src['shape']}, model expects {list(s.shape)}")
        full[s.offset:s.offset + s.numel].copy_(master[src["offset"]:src["offset"] + s.numel])
    strategy.import_master(full)
    bufs = dict(model.named_buffers())
    for n, b in ck["buffers"].items():
        if n in bufs:
            bufs[n].copy_(b.to(bufs[n].device))
    path = os.path.join(directory, f"optim_rank{comm.rank}.pt")
    if os.path.exists(path):
        st = torch.load(path, map_location="cpu", weights_only=True)
        opts = getattr(strategy, "optimizers", None) or ([strategy.optimizer] if strategy.optimizer is not None else [])
        for o, sd in zip(opts, st["optimizers"]):
            o.load_state_dict(sd)
        for m, sd in zip(getattr(strategy, "sparse_modules", []), st.get("sparse", [])):
            m.load_state_dict_shard(sd)
        if "rng" in st:
            _set_rng_state(st["rng"], comm.device)
    strategy.replica.step_count = int(ck["step"])
    comm.barrier()
    return int(ck["step"])
