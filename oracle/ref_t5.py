"""Oracle for T5 v1.1 (test infrastructure, see oracle/__init__.py): encoder, decoder and full model restated over a
state_dict in fp32.

Restates /root/reference pytorch_models/text/t5.py:15-25 (LayerNorm without centring or bias), :29-38 (GEGLU, tanh GELU),
:41-70 (relative-position buckets: exact below n/2, logarithmic to max_distance, clipped; the log in float32 with the
float32 epsilon added, truncated), :73-97 (pre-norm block: self-attention with the bias table, optional cross-attention,
gated MLP, all projections bias-free), :100-132 (encoder: bidirectional buckets; decoder: one-sided buckets + causal mask),
:135-151 (shared embedding, classifier) and :213-227 (greedy loop from the pad id).
PINNED: tests/golden/t5.npz holds the reference's own outputs on synthweights (tests/golden/make_golden.py t5).
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from . import ref_transformer as T

EPS = 1e-5
N_BUCKETS, MAX_DISTANCE = 32, 128


def rmsnorm(w: Tensor, x: Tensor) -> Tensor:
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + EPS) * w


def buckets(length: int, bidirectional: bool) -> Tensor:
    q = torch.arange(length)[:, None]
    k = torch.arange(length)[None, :]
    rel = k - q
    if bidirectional:
        n = N_BUCKETS // 2
        base = torch.where(rel > 0, n, 0)
        dist = rel.abs()
    else:
        n = N_BUCKETS
        base = torch.zeros_like(rel)
        dist = torch.where(rel < 0, -rel, 0)
    exact = n // 2
    log_bins = (n - exact) / math.log(MAX_DISTANCE / exact)
    far = exact + (torch.log(dist.to(torch.float32) / exact + torch.finfo(torch.float32).eps) * log_bins).to(torch.int64)
    far = torch.minimum(far, torch.tensor(n - 1))
    return torch.where(dist < exact, dist, far) + base


def attention(sd: dict, p: str, n_heads: int, x: Tensor, kv: Tensor, bias: Tensor | None) -> Tensor:
    """MHA(bias=False, head_dim=64) with an additive (H, Lq, Lk) bias - transformer.py:28-53."""
    q = T.split_heads(x @ sd[p + "q_proj.weight"].T, n_heads)
    k = T.split_heads(kv @ sd[p + "k_proj.weight"].T, n_heads)
    v = T.split_heads(kv @ sd[p + "v_proj.weight"].T, n_heads)
    return T.merge_heads(T.sdpa(q, k, v, attn_bias=bias)) @ sd[p + "out_proj.weight"].T


def block(sd: dict, p: str, n_heads: int, x: Tensor, memory: Tensor | None, bias: Tensor) -> Tensor:
    h = rmsnorm(sd[p + "sa_norm.weight"], x)
    x = x + attention(sd, p + "sa.", n_heads, h, h, bias)
    if memory is not None:
        x = x + attention(sd, p + "ca.", n_heads, rmsnorm(sd[p + "ca_norm.weight"], x), memory, None)
    h = rmsnorm(sd[p + "mlp_norm.weight"], x)
    g = T.activation(h @ sd[p + "mlp.0.w.weight"].T, "approximate_gelu") * (h @ sd[p + "mlp.0.v.weight"].T)
    return x + g @ sd[p + "mlp.2.weight"].T


def _stack(sd: dict, p: str, x: Tensor, memory: Tensor | None, bias: Tensor) -> Tensor:
    n_heads = sd[p + "attn_bias.bias"].shape[0]
    for i in range(T.n_layers_of(sd, p + "layers.")):
        x = block(sd, f"{p}layers.{i}.", n_heads, x, memory, bias)
    return rmsnorm(sd[p + "norm.weight"], x)


def encoder(sd: dict, p: str, x: Tensor) -> Tensor:
    bias = sd[p + "attn_bias.bias"][:, buckets(x.shape[-2], True)]
    return _stack(sd, p, x, None, bias)


def decoder(sd: dict, p: str, x: Tensor, memory: Tensor) -> Tensor:
    L = x.shape[-2]
    bias = sd[p + "attn_bias.bias"][:, buckets(L, False)] + torch.full((L, L), -1e10).triu(1)
    return _stack(sd, p, x, memory, bias)


def model(sd: dict, tokens: Tensor, targets: Tensor) -> Tensor:
    E = sd["token_embs.weight"]
    return decoder(sd, "decoder.", E[targets], encoder(sd, "encoder.", E[tokens])) @ sd["classifier.weight"].T


def greedy(sd: dict, tokens: Tensor, max_tokens: int, pad_id: int = 0, eos_id: int = 1):
    """T5Generator.generate on ids (t5.py:213-227), one sequence; also returns the top-2 margin of every decision."""
    E = sd["token_embs.weight"]
    memory = encoder(sd, "encoder.", E[tokens][None])
    out, margins = [pad_id], []
    while len(out) < max_tokens:
        lg = (decoder(sd, "decoder.", E[torch.tensor([out])], memory) @ sd["classifier.weight"].T)[0, -1]
        top2 = lg.topk(2).values
        margins.append((top2[0] - top2[1]).item())
        out.append(int(lg.argmax()))
        if out[-1] == eos_id:
            break
    return torch.tensor(out), margins
