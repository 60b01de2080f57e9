"""BERT encoder with a masked-language-model head (BERT-base by default).

The "BERT-base MLM, synthetic sequence length 512" benchmark configuration; the reference itself
has no sequence model (reference initializer.py:12-21). Every dense layer runs on the MFMA GEMM
with bias (+GELU) in the epilogue, attention on the batched GEMM + fused softmax, LayerNorm is
fused with the residual add, the three embedding tables are summed in one gather pass, and the
decoder (tied to the word embeddings) feeds the fused softmax-cross-entropy over the vocabulary.

Input ``x``: ``int64 [B, S]`` token ids, or ``[B, S + P]`` where the last ``P`` columns are the
positions to predict (the MLM head then scores only those, as the original pre-training does
with ``max_predictions_per_seq``). Output: logits ``[B*S or B*P, vocab]``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from ..api.model import Model
from ..ops import attention as attn_ops
from ..ops import dense as dense_ops
from ..ops import embedding as emb_ops
from ..ops import layernorm as ln_ops


class Linear(nn.Module):
    def __init__(self, cin: int, cout: int, activation: str | None = None, std: float = 0.02):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.activation = activation

    def forward(self, x):
        return dense_ops.linear(x, self.weight, self.bias, self.activation)

    def keras_weights(self):
        return [self.weight.detach().float().cpu().numpy().T.copy(), self.bias.detach().float().cpu().numpy()]

    def set_keras_weights(self, ws):
        with torch.no_grad():
            self.weight.copy_(torch.from_numpy(np.asarray(ws[0]).T.copy()).to(self.weight.dtype))
            self.bias.copy_(torch.from_numpy(np.asarray(ws[1])).to(self.bias.dtype))
        return 2


class LayerNorm(nn.Module):
    keep_fp32 = True

    def __init__(self, h: int, eps: float = 1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(h))
        self.bias = nn.Parameter(torch.zeros(h))
        self.eps = eps

    def forward(self, x, residual=None):
        return ln_ops.layer_norm(x, self.weight, self.bias, self.eps, residual)

    def keras_weights(self):
        return [self.weight.detach().float().cpu().numpy(), self.bias.detach().float().cpu().numpy()]

    def set_keras_weights(self, ws):
        with torch.no_grad():
            self.weight.copy_(torch.from_numpy(np.asarray(ws[0])).to(self.weight.dtype))
            self.bias.copy_(torch.from_numpy(np.asarray(ws[1])).to(self.bias.dtype))
        return 2


class EncoderLayer(nn.Module):
    def __init__(self, h: int, heads: int, ffn: int):
        super().__init__()
        self.heads = heads
        self.qkv = Linear(h, 3 * h)
        self.proj = Linear(h, h)
        self.ln1 = LayerNorm(h)
        self.ffn1 = Linear(h, ffn, activation="gelu")
        self.ffn2 = Linear(ffn, h)
        self.ln2 = LayerNorm(h)

    def forward(self, x, key_bias=None):
        B, S, H = x.shape
        # the aliases carry the residual branches' gradients into the data-gradient GEMM epilogues
        qkv, xr = dense_ops.linear_fork(x, self.qkv.weight, self.qkv.bias)
        ctxv = attn_ops.self_attention(qkv.view(B, S, 3, self.heads, H // self.heads), key_bias).view(B, S, H)
        x = self.ln1(self.proj(ctxv), residual=xr)
        f, xr = dense_ops.ffn_fork(x, self.ffn1.weight, self.ffn1.bias, self.ffn2.weight, self.ffn2.bias)
        return self.ln2(f, residual=xr)


class Bert(Model):
    supports_logits = True
    input_layout = "tokens"

    def __init__(self, vocab: int = 30522, hidden: int = 768, layers: int = 12, heads: int = 12, ffn: int = 3072,
                 max_pos: int = 512, type_vocab: int = 2, seq_len: int | None = None):
        super().__init__()
        self.vocab, self.hidden, self.seq_len = vocab, hidden, seq_len
        self.word = nn.Parameter(torch.empty(vocab, hidden).normal_(0, 0.02))
        self.pos = nn.Parameter(torch.empty(max_pos, hidden).normal_(0, 0.02))
        self.typ = nn.Parameter(torch.empty(type_vocab, hidden).normal_(0, 0.02))
        self.emb_ln = LayerNorm(hidden)
        self.layers = nn.ModuleList([EncoderLayer(hidden, heads, ffn) for _ in range(layers)])
        self.head_dense = Linear(hidden, hidden, activation="gelu")
        self.head_ln = LayerNorm(hidden)
        self.decoder_bias = nn.Parameter(torch.zeros(vocab))

    def prepare_input(self, x):
        return x

    def forward(self, x, logits: bool = False):
        S = self.seq_len if (self.seq_len is not None and x.shape[1] > self.seq_len) else x.shape[1]
        ids, positions = x[:, :S], (x[:, S:] if x.shape[1] > S else None)
        B = ids.shape[0]
        h = emb_ops.embedding(self.word, ids, self.pos, self.typ)
        h = self.emb_ln(h)
        for l in self.layers:
            h = l(h)
        if positions is not None:
            idx = (positions + torch.arange(B, device=x.device)[:, None] * S).reshape(-1)
            h = h.reshape(B * S, -1).index_select(0, idx)
        else:
            h = h.reshape(B * S, -1)
        h = self.head_ln(self.head_dense(h))
        out = dense_ops.linear(h, self.word, self.decoder_bias)    # decoder tied to the word embeddings
        return out if logits else torch.softmax(out.float(), dim=-1)


def bert_base(seq_len: int = 512, **kw) -> Bert:
    return Bert(seq_len=seq_len, **kw)


def bert_tiny(seq_len: int = 64, **kw) -> Bert:
    kw = dict(dict(vocab=1024, hidden=128, layers=2, heads=2, ffn=512, max_pos=128), **kw)
    return Bert(seq_len=seq_len, **kw)


def synthetic_mlm_batch(batch: int, seq_len: int, vocab: int, device, n_pred: int = 80, seed: int = 0):
    """``(x, y)``: ``x = [ids | positions]`` of shape ``[B, S + P]``, ``y = labels [B*P]`` (random tokens)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    ids = torch.randint(0, vocab, (batch, seq_len), generator=g, dtype=torch.int64)
    pos = torch.stack([torch.randperm(seq_len, generator=g)[:n_pred].sort().values for _ in range(batch)])
    labels = ids.gather(1, pos).clone()
    ids.scatter_(1, pos, 103 % vocab)   # [MASK]
    return torch.cat([ids, pos], dim=1).to(device), labels.reshape(-1).to(device)
