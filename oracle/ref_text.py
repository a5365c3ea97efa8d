"""Oracle for the text models (test infrastructure, see oracle/__init__.py): GPT-2, GPT and BERT forwards and the
greedy loop, restated over a state_dict in fp32.

Restates /root/reference pytorch_models/text/gpt2.py:14-27 (pre-norm causal decoder, tanh-GELU, final LayerNorm, logits
against the tied embedding), gpt.py:15-29 (post-norm, no final norm), bert.py:17-40 (embeddings -> LayerNorm(eps 1e-12)
-> post-norm encoder, exact GELU) and generator.py:23-35 (greedy: argmax of the last position, append, repeat).
PINNED: tests/golden/text.npz holds the reference's own outputs on synthweights (tests/golden/make_golden.py text).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import ref_transformer as T

HEAD_DIM = 64  # MHA default - transformer.py:20-22
EPS = 1e-5     # nn.LayerNorm / DecoderLayer default


def _embed(sd: dict, tokens: Tensor) -> Tensor:
    return sd["token_embs.weight"][tokens] + sd["pos_embs"][: tokens.shape[-1]]


def gpt2(sd: dict, tokens: Tensor, rp=None) -> Tensor:
    d = sd["pos_embs"].shape[1]
    x = T.decoder(sd, "layers.", d // HEAD_DIM, _embed(sd, tokens), None, act="approximate_gelu", rp=rp)
    return T.layernorm(sd, "norm.", x, EPS) @ sd["token_embs.weight"].T


def gpt(sd: dict, tokens: Tensor) -> Tensor:
    d = sd["pos_embs"].shape[1]
    x = T.decoder(sd, "layers.", d // HEAD_DIM, _embed(sd, tokens), None, act="approximate_gelu", pre_norm=False)
    return x @ sd["token_embs.weight"].T


def bert(sd: dict, tokens: Tensor, eps: float = 1e-12) -> Tensor:
    d = sd["pos_embs"].shape[1]
    x = T.layernorm(sd, "norm.", _embed(sd, tokens), eps)
    return T.encoder(sd, "layers.", d // HEAD_DIM, x, pre_norm=False, eps=eps)


def greedy(forward, sd: dict, prompt: Tensor, n_new: int, rp=None):
    """generator.py:23-35 for a batch: full-prefix recompute per token.  Returns (ids (B, P + n_new), margins (B, n_new)
    = top1 - top2 logit gap at every decided position, for classifying near-ties in tests)."""
    toks = prompt.clone()
    margins = []
    for _ in range(n_new):
        logits = (forward(sd, toks, rp=rp) if rp is not None else forward(sd, toks))[:, -1]
        top2 = logits.topk(2, -1).values
        margins.append(top2[:, 0] - top2[:, 1])
        toks = torch.cat([toks, logits.argmax(-1, keepdim=True)], 1)
    return toks, torch.stack(margins, 1)
