"""Oracle for Whisper (test infrastructure, see oracle/__init__.py).

Restates /root/reference pytorch_models/audio2text/whisper.py:11-94.  The greedy loop has no
Whisper counterpart in the reference (README.md:86); its semantics follow the reference's generic
generators (text/generator.py:23-35, text/t5.py:219-225): argmax of the last position's logits,
append, repeat - by full-prefix recompute (``greedy_recompute``), with a KV-cached equivalent
(``greedy_cached``) that tests prove identical.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import ref_transformer as T

# tag -> (n_layers, d_model) - whisper.py:67-79 ("base" is 8 layers in the reference: SURVEY.md F2)
SIZES = {
    "tiny": (4, 384), "tiny.en": (4, 384), "base": (8, 512), "base.en": (8, 512),
    "small": (12, 768), "small.en": (12, 768), "medium": (24, 1024), "medium.en": (24, 1024),
    "large-v1": (32, 1280), "large-v2": (32, 1280), "large-v3": (32, 1280),
}
EPS = 1e-5  # nn.LayerNorm default, DecoderLayer default - transformer.py:82
HEAD_DIM = 64  # MHA default when neither n_heads nor head_dim is given - transformer.py:20-22


def geometry_from_openai(tag: str) -> dict:
    """whisper.py:67-86: large-v3 => 128 mels, vocab 51866; *.en => vocab 51864; else 51865."""
    n_layers, d = SIZES[tag]
    if tag == "large-v3":
        n_mels, vocab = 128, 51866
    else:
        n_mels, vocab = 80, (51864 if tag.endswith(".en") else 51865)
    return dict(vocab_size=vocab, n_layers=n_layers, d_model=d, n_mels=n_mels)


def conv1d(sd: dict, p: str, x: Tensor, stride: int) -> Tensor:
    """nn.Conv1d(cin, cout, 3, stride, padding=1) as an explicit (cin*3)-deep contraction - whisper.py:17,19."""
    w = sd[p + "weight"]  # (cout, cin, 3)
    xp = torch.nn.functional.pad(x, (1, 1))
    cols = xp.unfold(-1, 3, stride)  # (B, cin, T_out, 3)
    return torch.einsum("bctk,ock->bot", cols, w) + sd[p + "bias"][:, None]


def encoder(sd: dict, p: str, x: Tensor, rp=None) -> Tensor:
    """WhisperEncoder.forward - whisper.py:29-34.  x: (B, n_mels, T) -> (B, T/2, d)."""
    d = sd[p + "stem.0.weight"].shape[0]
    x = T.activation(conv1d(sd, p + "stem.0.", x, 1), "gelu")
    if rp is not None:
        x = rp("stem", x)
    x = T.activation(conv1d(sd, p + "stem.2.", x, 2), "gelu").transpose(1, 2)
    x = x + sd[p + "pos_embs"][: x.shape[1]]
    if rp is not None:
        x = rp("resid", x)
    x = T.encoder(sd, p + "layers.", d // HEAD_DIM, x, eps=EPS, rp=rp)
    x = T.layernorm(sd, p + "norm.", x, EPS)
    return x if rp is None else rp("memory", x)


def decoder(sd: dict, p: str, tokens: Tensor, memory: Tensor, rp=None) -> Tensor:
    """WhisperDecoder.forward - whisper.py:47-53: logits (B, L, V) with tied embeddings."""
    E = sd[p + "token_embs.weight"]
    d = E.shape[1]
    x = E[tokens] + sd[p + "pos_embs"][: tokens.shape[1]]
    x = T.decoder(sd, p + "layers.", d // HEAD_DIM, x, memory, eps=EPS, rp=rp)
    x = T.layernorm(sd, p + "norm.", x, EPS)
    return x @ E.T


def forward(sd: dict, mel: Tensor, targets: Tensor) -> Tensor:
    """Whisper.forward - whisper.py:62-63."""
    return decoder(sd, "decoder.", targets, encoder(sd, "encoder.", mel))


@torch.no_grad()
def greedy_recompute(sd: dict, p: str, prompt: Tensor, memory: Tensor, n_new: int):
    """Reference-semantics greedy decode: every step re-runs the decoder on the whole prefix.
    Returns (tokens (B, P + n_new) int64, margins (B, n_new) fp32 = top1 - top2 logit)."""
    toks = prompt.clone()
    margins = []
    for _ in range(n_new):
        last = decoder(sd, p, toks, memory)[:, -1]
        top2 = last.topk(2, dim=-1)
        margins.append(top2.values[:, 0] - top2.values[:, 1])
        toks = torch.cat([toks, top2.indices[:, :1]], dim=1)
    return toks, torch.stack(margins, 1)


@torch.no_grad()
def greedy_cached(sd: dict, p: str, prompt: Tensor, memory: Tensor, n_new: int, rp=None):
    """KV-cached greedy decode, algebraically equal to ``greedy_recompute``: causal prefill of the
    prompt, then one token per step attending to the cached self K/V (NO causal flag: with L_q = 1 a
    top-left-aligned mask would hide all but key 0 - SURVEY.md F3) and to cross K/V projected once.

    ``rp`` marks the HIP path's storage points ("kv" for cached K/V, "memory")."""
    E = sd[p + "token_embs.weight"]
    d = E.shape[1]
    h = d // HEAD_DIM
    B, P = prompt.shape
    L = T.n_layers_of(sd, p + "layers.")
    r = (lambda n, t: t) if rp is None else rp
    cross = []
    for i in range(L):
        q = p + f"layers.{i}.ca."
        cross.append((T.split_heads(r("kv", T.linear(sd, q + "k_proj.", memory)), h),
                      T.split_heads(r("kv", T.linear(sd, q + "v_proj.", memory)), h)))
    self_k = [None] * L
    self_v = [None] * L

    def step(tok: Tensor, pos0: int, causal: bool) -> Tensor:
        x = E[tok] + sd[p + "pos_embs"][pos0 : pos0 + tok.shape[1]]
        for i in range(L):
            lp = p + f"layers.{i}."
            xn = T.layernorm(sd, lp + "sa_norm.", x, EPS)
            q = T.split_heads(T.linear(sd, lp + "sa.q_proj.", xn), h)
            k = T.split_heads(r("kv", T.linear(sd, lp + "sa.k_proj.", xn)), h)
            v = T.split_heads(r("kv", T.linear(sd, lp + "sa.v_proj.", xn)), h)
            self_k[i] = k if self_k[i] is None else torch.cat([self_k[i], k], dim=-2)
            self_v[i] = v if self_v[i] is None else torch.cat([self_v[i], v], dim=-2)
            a = T.sdpa(q, self_k[i], self_v[i], causal=causal)
            x = x + T.linear(sd, lp + "sa.out_proj.", T.merge_heads(a))
            xn = T.layernorm(sd, lp + "ca_norm.", x, EPS)
            q = T.split_heads(T.linear(sd, lp + "ca.q_proj.", xn), h)
            a = T.sdpa(q, cross[i][0], cross[i][1])
            x = x + T.linear(sd, lp + "ca.out_proj.", T.merge_heads(a))
            x = x + T.mlp(sd, lp + "mlp.", T.layernorm(sd, lp + "mlp_norm.", x, EPS))
        x = T.layernorm(sd, p + "norm.", x[:, -1:], EPS)
        return (x @ E.T)[:, 0]

    toks = prompt.clone()
    margins = []
    last = step(prompt, 0, causal=True)
    for t in range(n_new):
        top2 = last.topk(2, dim=-1)
        margins.append(top2.values[:, 0] - top2.values[:, 1])
        nxt = top2.indices[:, :1]
        toks = torch.cat([toks, nxt], dim=1)
        if t + 1 < n_new:
            last = step(nxt, P + t, causal=False)
    return toks, torch.stack(margins, 1)
