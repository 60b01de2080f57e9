"""Wide & Deep recommender with row-sharded embedding tables (parameter-server state).

The "Wide&Deep, 200 GB sharded embedding tables, centralized sync PS" configuration. The model
follows Cheng et al. 2016: a *wide* linear part over sparse (crossed) features and a *deep* MLP
over dense features concatenated with embeddings of the sparse features, summed into one
logit trained with binary cross-entropy. All sparse features index ONE hashed id space that is
row-sharded over the GPUs (``parallel/sharded_embedding.py``); the dense MLP is replicated and
goes through the flat-arena / collective path like any other model.

Batch layout: ``x = (ids int64 [B, F], dense float [B, n_dense])``, ``y`` float ``[B]`` in {0, 1}.
"""
from __future__ import annotations

import json
import time

import torch
import torch.nn as nn

from ..api.model import Model
from ..comm.backend import Comm
from ..ops import dense as dense_ops
from ..parallel.sharded_embedding import ShardedEmbedding


class _Dense(nn.Module):
    def __init__(self, cin, cout, act=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        nn.init.xavier_uniform_(self.weight)
        self.bias = nn.Parameter(torch.zeros(cout))
        self.act = act

    def forward(self, x):
        return dense_ops.linear(x, self.weight, self.bias, self.act)


class WideDeep(Model):
    supports_logits = True
    input_layout = "features"

    def __init__(self, comm: Comm, num_rows: int, n_sparse: int = 26, n_dense: int = 13, emb_dim: int = 128,
                 hidden=(1024, 512, 256), wide_dim: int = 4, emb_optimizer: str = "adagrad", emb_lr: float = 0.05,
                 init: str = "normal"):
        super().__init__()
        self.n_sparse, self.n_dense, self.emb_dim = n_sparse, n_dense, emb_dim
        self.deep_emb = ShardedEmbedding(num_rows, emb_dim, comm, emb_optimizer, emb_lr, init=init)
        self.wide_emb = ShardedEmbedding(num_rows, wide_dim, comm, emb_optimizer, emb_lr, init="zeros")
        dims = [n_sparse * emb_dim + _pad8(n_dense), *hidden]
        self.mlp = nn.ModuleList([_Dense(a, b, "relu") for a, b in zip(dims[:-1], dims[1:])])
        self.out = _Dense(dims[-1], 8)          # 8 output columns keep rows 16-byte aligned; column 0 is the logit
        self.wide_dense = _Dense(_pad8(n_dense), 8)

    def prepare_input(self, x):
        return x

    def sparse_modules(self):
        return [self.deep_emb, self.wide_emb]

    def forward(self, x, logits: bool = False):
        ids, dense = x
        B = ids.shape[0]
        wdt = self.out.weight.dtype
        dense = torch.nn.functional.pad(dense.to(wdt), (0, _pad8(self.n_dense) - self.n_dense))
        plan = self.deep_emb.plan(ids)           # both tables are indexed by the same ids: one routing plan for the two
        emb = self.deep_emb(ids, plan).to(wdt).reshape(B, -1)
        h = torch.cat([emb, dense], dim=1)
        for l in self.mlp:
            h = l(h)
        deep = self.out(h)[:, 0].float()
        wide = self.wide_emb(ids, plan).float().sum(dim=(1, 2)) + self.wide_dense(dense)[:, 0].float()
        z = deep + wide
        return z if logits else torch.sigmoid(z)


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def bce_loss(model, x, y):
    """Per-example binary cross-entropy on the logit (numerically stable form)."""
    z = model(x, logits=True)
    y = y.float()
    return torch.clamp(z, min=0) - z * y + torch.log1p(torch.exp(-z.abs()))


def synthetic_batch(batch: int, num_rows: int, n_sparse: int, n_dense: int, device, seed: int = 0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    # skewed ids (a few hot rows, a long tail), as in click logs
    u = torch.rand(batch, n_sparse, generator=g)
    ids = (u.pow(3) * (num_rows - 1)).long()
    dense = torch.randn(batch, n_dense, generator=g)
    y = (torch.rand(batch, generator=g) < 0.25).float()
    return (ids.to(device), dense.to(device)), y.to(device)


def bench_step(device, steps: int = 10, warmup: int = 3, table_gb: float = 25.0, batch: int = 8192) -> dict:
    """Single-GPU slice of the configuration: one shard (``table_gb`` of rows; 200 GB / 8 GPUs = 25 GB
    per GPU) plus the replicated MLP, synchronous parameter-server step."""
    from ..engine.optimizer import Adam
    from ..engine.replica import TrainConfig
    from ..parallel.ps import SyncPSStrategy
    comm = Comm(0, 1, device)
    emb_dim = 128
    rows = int(table_gb * (1 << 30) / (emb_dim * 4))
    torch.manual_seed(0)
    model = WideDeep(comm, rows, emb_dim=emb_dim)
    cfg = TrainConfig(loss=bce_loss, loss_reduction="mean", param_dtype=torch.bfloat16)
    st = SyncPSStrategy(model, comm, cfg, optimizer=Adam(learning_rate=1e-3))
    x, y = synthetic_batch(batch, rows, model.n_sparse, model.n_dense, device)
    for _ in range(warmup):
        st.train_step(x, y)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = st.train_step(x, y)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    res = {"model": f"wide_deep-{table_gb:g}GB-shard-b{batch}", "ms_per_step": round(dt * 1e3, 3),
           "examples_per_s": round(batch / dt, 1), "loss": float(loss),
           "ps_state_gb": round(sum(m.nbytes() for m in model.sparse_modules()) / (1 << 30), 2)}
    print(json.dumps(res), flush=True)
    return res
