"""Oracle for the shared transformer blocks (test infrastructure, see oracle/__init__.py).

Restates /root/reference pytorch_models/transformer.py in plain fp32 tensor algebra:
explicit matmul + softmax instead of ``F.scaled_dot_product_attention``, explicit mean/variance
instead of ``nn.LayerNorm``, explicit erf instead of ``nn.GELU``.  Each function cites the
reference lines it follows.  ``sd`` is a state_dict, ``p`` a key prefix ending in "." or "".
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
from torch import Tensor

Round = Optional[Callable[[str, Tensor], Tensor]]


def _r(rp: Round, name: str, x: Tensor) -> Tensor:
    """Optional rounding point: tests that mirror the HIP path's bf16 storage points pass a hook."""
    return x if rp is None else rp(name, x)


def linear(sd: dict, p: str, x: Tensor) -> Tensor:
    """nn.Linear: y = x W^T + b with W stored (out, in) - transformer.py:28-31."""
    y = x @ sd[p + "weight"].T
    b = sd.get(p + "bias")
    return y if b is None else y + b


def layernorm(sd: dict, p: str, x: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance - transformer.py:87,90,93."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * sd[p + "weight"] + sd[p + "bias"]


def activation(x: Tensor, act: str) -> Tensor:
    """Activation table of MLP - transformer.py:60-65."""
    if act == "gelu":
        return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))
    if act == "approximate_gelu":
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x * x * x)))
    if act == "relu":
        return x.clamp_min(0)
    if act == "silu":
        return x * torch.sigmoid(x)
    raise KeyError(act)


def sdpa(q: Tensor, k: Tensor, v: Tensor, attn_bias: Tensor | None = None, causal: bool = False) -> Tensor:
    """softmax(q k^T / sqrt(hd) + bias [+ causal]) v for (*, h, L, hd) operands - transformer.py:52.

    ``causal`` is TOP-LEFT aligned like torch's ``is_causal=True``: query i sees keys j <= i even
    when L_q != S_k (SURVEY.md finding F3)."""
    hd = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if attn_bias is not None:
        if attn_bias.dtype == torch.bool:
            s = s.masked_fill(~attn_bias, float("-inf"))
        else:
            s = s + attn_bias
    if causal:
        L, S = s.shape[-2], s.shape[-1]
        keep = torch.ones(L, S, dtype=torch.bool, device=s.device).tril()
        s = s.masked_fill(~keep, float("-inf"))
    s = s - s.amax(-1, keepdim=True)
    w = torch.exp(s)
    w = w / w.sum(-1, keepdim=True)
    return w @ v


def split_heads(x: Tensor, n_heads: int) -> Tensor:
    """(*, L, h*hd) -> (*, h, L, hd) - transformer.py:47."""
    return x.unflatten(-1, (n_heads, x.shape[-1] // n_heads)).transpose(-2, -3)


def merge_heads(x: Tensor) -> Tensor:
    """(*, h, L, hd) -> (*, L, h*hd) - transformer.py:53."""
    return x.transpose(-2, -3).flatten(-2)


def mha(
    sd: dict,
    p: str,
    n_heads: int,
    q: Tensor,
    k: Tensor | None = None,
    v: Tensor | None = None,
    attn_bias: Tensor | None = None,
    causal: bool = False,
    rp: Round = None,
) -> Tensor:
    """MHA.forward - transformer.py:36-53.  k defaults to q and v to k (:44-45)."""
    k = q if k is None else k
    v = k if v is None else v
    qh = split_heads(_r(rp, "q", linear(sd, p + "q_proj.", q)), n_heads)
    kh = split_heads(_r(rp, "k", linear(sd, p + "k_proj.", k)), n_heads)
    vh = split_heads(_r(rp, "v", linear(sd, p + "v_proj.", v)), n_heads)
    out = merge_heads(sdpa(qh, kh, vh, attn_bias, causal))
    return linear(sd, p + "out_proj.", _r(rp, "attn_out", out))


def mlp(sd: dict, p: str, x: Tensor, act: str = "gelu", rp: Round = None) -> Tensor:
    """MLP: linear1 -> act -> linear2 (dropout is identity in eval) - transformer.py:56-67."""
    h = _r(rp, "mlp_hidden", activation(linear(sd, p + "linear1.", x), act))
    return linear(sd, p + "linear2.", h)


def decoder_layer(
    sd: dict,
    p: str,
    n_heads: int,
    x: Tensor,
    memory: Tensor | None = None,
    *,
    pre_norm: bool = True,
    eps: float = 1e-5,
    act: str = "gelu",
    causal: bool = True,
    rp: Round = None,
) -> Tensor:
    """DecoderLayer.forward - transformer.py:96-105; ``causal=False`` gives EncoderLayer.forward
    (transformer.py:123-130).  Cross-attention exists iff the state_dict has ``ca`` weights."""
    has_ca = (p + "ca.q_proj.weight") in sd
    if pre_norm:
        x = x + mha(sd, p + "sa.", n_heads, _r(rp, "ln", layernorm(sd, p + "sa_norm.", x, eps)), causal=causal, rp=rp)
        x = _r(rp, "resid", x)
        if has_ca:
            x = x + mha(sd, p + "ca.", n_heads, _r(rp, "ln", layernorm(sd, p + "ca_norm.", x, eps)), memory, rp=rp)
            x = _r(rp, "resid", x)
        x = x + mlp(sd, p + "mlp.", _r(rp, "ln", layernorm(sd, p + "mlp_norm.", x, eps)), act, rp=rp)
        x = _r(rp, "resid", x)
    else:
        x = layernorm(sd, p + "sa_norm.", x + mha(sd, p + "sa.", n_heads, x, causal=causal, rp=rp), eps)
        if has_ca:
            x = layernorm(sd, p + "ca_norm.", x + mha(sd, p + "ca.", n_heads, x, memory, rp=rp), eps)
        x = layernorm(sd, p + "mlp_norm.", x + mlp(sd, p + "mlp.", x, act, rp=rp), eps)
    return x


def encoder_layer(sd: dict, p: str, n_heads: int, x: Tensor, **kw) -> Tensor:
    """EncoderLayer.forward - transformer.py:123-130."""
    return decoder_layer(sd, p, n_heads, x, None, causal=False, **kw)


def n_layers_of(sd: dict, p: str) -> int:
    n = 0
    while (p + f"{n}.sa.q_proj.weight") in sd:
        n += 1
    return n


def encoder(sd: dict, p: str, n_heads: int, x: Tensor, **kw) -> Tensor:
    """Encoder (nn.Sequential of EncoderLayer) - transformer.py:133-149."""
    for i in range(n_layers_of(sd, p)):
        x = encoder_layer(sd, p + f"{i}.", n_heads, x, **kw)
    return x


def decoder(sd: dict, p: str, n_heads: int, x: Tensor, memory: Tensor | None = None, **kw) -> Tensor:
    """Decoder.forward - transformer.py:173-176."""
    for i in range(n_layers_of(sd, p)):
        x = decoder_layer(sd, p + f"{i}.", n_heads, x, memory, **kw)
    return x
