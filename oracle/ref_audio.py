"""Oracle for the wav2vec2 family (test infrastructure, see oracle/__init__.py): Wav2Vec2 / HuBERT, data2vec-audio and
SEW forwards restated over a state_dict in fp32, channel-LAST throughout (explicit window contractions, explicit
statistics, no nn.Conv1d / nn.LayerNorm / nn.InstanceNorm1d).

Restates /root/reference pytorch_models/audio/wav2vec2.py:19-39 (feature encoder: conv -> LayerNorm1d over channels |
InstanceNorm1d(affine) over time on layer 0 only | nothing -> GELU), :66-83 (LayerNorm [+ Linear] projection, x +
GELU(grouped conv(pad(x))), pre- or post-norm encoder), audio/data2vec_audio.py:23-30 (five grouped conv -> LayerNorm
without affine -> GELU layers) and audio/sew.py:27-39 (avg-pool + stride-2 positional conv, up-sampling Linear + GELU,
zero frame for odd lengths).
PINNED: tests/golden/audio_enc.npz holds the reference's own outputs on synthweights (tests/golden/make_golden.py
audio_enc).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import ref_transformer as T

HEAD_DIM = 64
EPS = 1e-5
PE_GROUPS = 16  # Wav2Vec2.PE_GROUPS - wav2vec2.py:49


def conv1d_tl(w: Tensor, b: Tensor | None, x: Tensor, stride: int, groups: int = 1) -> Tensor:
    """Conv1d on channel-last x (B, T, C): y[b, t, o] = b[o] + sum_{c, j} w[o, c, j] x[b, t*stride + j, g(o)*cg + c]."""
    co, cg, k = w.shape
    cols = x.unfold(1, k, stride)  # (B, To, C, k)
    B, To = cols.shape[:2]
    cols = cols.reshape(B, To, groups, cg, k)
    y = torch.einsum("btgck,gock->btgo", cols, w.view(groups, co // groups, cg, k)).reshape(B, To, co)
    return y if b is None else y + b


def _norm_free(x: Tensor, dim: int, eps: float) -> Tensor:
    mu = x.mean(dim, keepdim=True)
    xc = x - mu
    return xc * torch.rsqrt((xc * xc).mean(dim, keepdim=True) + eps)


def feature_encoder(sd: dict, p: str, x: Tensor, strides, legacy: bool) -> Tensor:
    """wav2vec2.py:19-39 on a waveform (B, L) -> (B, T, C) channel-last."""
    h = x[:, :, None]
    for i, s in enumerate(strides):
        q = f"{p}{i}."
        h = conv1d_tl(sd[q + "0.weight"], sd.get(q + "0.bias"), h, s)
        if q + "2.weight" in sd:
            # legacy layer 0: InstanceNorm1d = per (clip, channel) statistics over time; else LayerNorm over channels
            h = _norm_free(h, 1 if legacy else 2, EPS) * sd[q + "2.weight"] + sd[q + "2.bias"]
        h = T.activation(h, "gelu")
    return h


def _project(sd: dict, h: Tensor) -> Tensor:
    h = T.layernorm(sd, "proj.0.", h, EPS)
    return T.linear(sd, "proj.1.", h) if "proj.1.weight" in sd else h


def _pad_time(h: Tensor, left: int, right: int) -> Tensor:
    B, _, d = h.shape
    return torch.cat([h.new_zeros(B, left, d), h, h.new_zeros(B, right, d)], 1)


STEM_STRIDES = (5,) + (2,) * 6


def wav2vec2(sd: dict, x: Tensor, *, pre_norm: bool = True, legacy: bool = False) -> Tensor:
    """Wav2Vec2.forward - wav2vec2.py:78-85."""
    h = _project(sd, feature_encoder(sd, "feature_encoder.", x, STEM_STRIDES, legacy))
    d = h.shape[-1]
    k = sd["pe_conv.1.weight"].shape[-1]
    pe = conv1d_tl(sd["pe_conv.1.weight"], sd["pe_conv.1.bias"], _pad_time(h, k // 2, k // 2 - 1), 1, PE_GROUPS)
    h = h + T.activation(pe, "gelu")
    if pre_norm:
        return T.layernorm(sd, "norm.", T.encoder(sd, "layers.", d // HEAD_DIM, h, eps=EPS), EPS)
    return T.encoder(sd, "layers.", d // HEAD_DIM, T.layernorm(sd, "norm.", h, EPS), pre_norm=False, eps=EPS)


def data2vec_audio(sd: dict, x: Tensor) -> Tensor:
    """Data2VecAudio - data2vec_audio.py:14-35 with the inherited forward."""
    h0 = _project(sd, feature_encoder(sd, "feature_encoder.", x, STEM_STRIDES, False))
    d = h0.shape[-1]
    h = h0
    i = 0
    while f"pe_conv.{i}.0.weight" in sd:
        w = sd[f"pe_conv.{i}.0.weight"]
        k = w.shape[-1]
        h = conv1d_tl(w, sd[f"pe_conv.{i}.0.bias"], _pad_time(h, k // 2, k // 2), 1, PE_GROUPS)
        h = T.activation(_norm_free(h, 2, EPS), "gelu")
        i += 1
    h = T.layernorm(sd, "norm.", h0 + h, EPS)
    return T.encoder(sd, "layers.", d // HEAD_DIM, h, pre_norm=False, eps=EPS)


SEW_STRIDES = (5,) + (2, 1) * 6


def sew(sd: dict, x: Tensor) -> Tensor:
    """SEW.forward - sew.py:27-39."""
    h = _project(sd, feature_encoder(sd, "feature_encoder.", x, SEW_STRIDES, True))
    B, Tn, d = h.shape
    k = sd["pe_conv.1.weight"].shape[-1]
    pe = T.activation(conv1d_tl(sd["pe_conv.1.weight"], sd["pe_conv.1.bias"], _pad_time(h, k // 2, k // 2 - 1), 2, PE_GROUPS), "gelu")
    pooled = h[:, : Tn // 2 * 2].reshape(B, Tn // 2, 2, d).mean(2)
    h = T.encoder(sd, "layers.", d // HEAD_DIM, T.layernorm(sd, "norm.", pooled + pe, EPS), pre_norm=False, eps=EPS)
    up = T.activation(T.linear(sd, "upsample.0.", h), "gelu").reshape(B, -1, d)  # (B, T/2, 2d) -> (B, T/2 * 2, d)
    if up.shape[1] < Tn:
        up = torch.cat([up, up.new_zeros(B, Tn - up.shape[1], d)], 1)
    return up
