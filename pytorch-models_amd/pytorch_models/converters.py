"""Weight converters from upstream checkpoint formats into this build's (reference-named) parameters.

Same behaviour as the reference's loaders - vision_transformer / big_vision Flax ``.npz``
(/root/reference pytorch_models/image/vit.py:151-200,309-335), facebook / timm ViT state_dicts with fused qkv and
folded layer scale (vit.py:257-306), OpenAI Whisper state_dicts (audio2text/whisper.py:96-135) - except that
nothing is downloaded: callers pass a local file or an in-memory mapping.  Parameters are updated in place with
``copy_`` (which bumps ``_version``, so the packed / re-typed weight caches of the HIP path rebuild themselves).
"""
from __future__ import annotations

import os
from typing import Mapping

import numpy as np
import torch
from torch import Tensor, nn


def _as_tensors(src, prefix: str = "") -> dict[str, Tensor]:
    if isinstance(src, (str, os.PathLike)):
        if not os.path.exists(src):
            raise FileNotFoundError(f"{src}: this build does not download checkpoints; pass a local .npz / state_dict")
        src = np.load(src, allow_pickle=False)
    out = {}
    for k in (src.files if hasattr(src, "files") else src.keys()):
        if k.startswith(prefix):
            v = src[k]
            out[k[len(prefix):]] = v if isinstance(v, Tensor) else torch.from_numpy(np.asarray(v))
    return out


def _put(param: Tensor, value: Tensor) -> None:
    param.copy_(value.reshape(param.shape) if value.numel() == param.numel() and value.shape != param.shape else value)


def _flax_dense(lin: nn.Linear, w: dict, key: str) -> None:
    """Flax Dense / DenseGeneral: kernel is (in..., out...) -> nn.Linear (out, in)."""
    out_f, in_f = lin.weight.shape
    lin.weight.copy_(w.pop(key + "/kernel").reshape(in_f, out_f).T)
    if lin.bias is not None:
        lin.bias.copy_(w.pop(key + "/bias").reshape(-1))


def _flax_norm(ln: nn.LayerNorm, w: dict, key: str) -> None:
    ln.weight.copy_(w.pop(key + "/scale"))
    ln.bias.copy_(w.pop(key + "/bias"))


def _flax_attention(mha, w: dict, key: str) -> None:
    for ours, theirs in (("q_proj", "query"), ("k_proj", "key"), ("v_proj", "value"), ("out_proj", "out")):
        _flax_dense(getattr(mha, ours), w, f"{key}/{theirs}")


@torch.no_grad()
def load_flax_vit(vit, ckpt, *, big_vision: bool = False, prefix: str = "") -> list[str]:
    """ckpt: local ``.npz`` path or mapping.  Returns the unconsumed keys (the reference prints them)."""
    w = _as_tensors(ckpt, prefix)
    if big_vision:  # github.com/google-research/big_vision
        ln1, att, ln2, mlp = "LayerNorm_0", "MultiHeadDotProductAttention_0", "LayerNorm_1", "MlpBlock_0"
    else:  # github.com/google-research/vision_transformer
        ln1, att, ln2, mlp = "LayerNorm_0", "MultiHeadDotProductAttention_1", "LayerNorm_2", "MlpBlock_3"
    if vit.cls_token is not None:
        vit.cls_token.copy_(w.pop("cls"))
    if big_vision:
        vit.pe.copy_(w.pop("pos_embedding"))
    else:  # the position table has a slot for the cls token: fold it into cls_token
        pe = w.pop("Transformer/posembed_input/pos_embedding")
        vit.cls_token.add_(pe[:, 0])
        vit.pe.copy_(pe[:, 1:])
    vit.patch_embed.weight.copy_(w.pop("embedding/kernel").permute(3, 2, 0, 1))  # (P, P, 3, d) -> (d, 3, P, P)
    if vit.patch_embed.bias is not None:
        vit.patch_embed.bias.copy_(w.pop("embedding/bias"))
    _flax_norm(vit.norm, w, "Transformer/encoder_norm")
    for i, layer in enumerate(vit.layers):
        blk = f"Transformer/encoderblock_{i}"
        _flax_norm(layer.sa_norm, w, f"{blk}/{ln1}")
        _flax_attention(layer.sa, w, f"{blk}/{att}")
        _flax_norm(layer.mlp_norm, w, f"{blk}/{ln2}")
        _flax_dense(layer.mlp.linear1, w, f"{blk}/{mlp}/Dense_0")
        _flax_dense(layer.mlp.linear2, w, f"{blk}/{mlp}/Dense_1")
    pool = vit.pooler
    if hasattr(pool, "probe"):  # MAP head (big_vision only)
        pool.probe.copy_(w.pop("MAPHead_0/probe"))
        _flax_attention(pool.attn, w, "MAPHead_0/MultiHeadDotProductAttention_0")
        _flax_norm(pool.norm, w, "MAPHead_0/LayerNorm_0")
        _flax_dense(pool.mlp.linear1, w, "MAPHead_0/MlpBlock_0/Dense_0")
        _flax_dense(pool.mlp.linear2, w, "MAPHead_0/MlpBlock_0/Dense_1")
    return sorted(w)


@torch.no_grad()
def load_facebook_vit(vit, state_dict: Mapping[str, Tensor]) -> list[str]:
    """DeiT-3 / DINO / DINOv2 (timm-style) state_dict: fused qkv is split, layer scale (gamma_1/2 or ls1/2.gamma)
    is folded into out_proj / linear2."""
    w = dict(state_dict)

    def wb(mod, key):
        _put(mod.weight, w.pop(key + ".weight"))
        mod.bias.copy_(w.pop(key + ".bias"))

    wb(vit.patch_embed, "patch_embed.proj")
    pe = w.pop("pos_embed")
    n = vit.pe.shape[1]
    vit.pe.copy_(pe[:, -n:])
    vit.cls_token.copy_(w.pop("cls_token"))
    if pe.shape[1] > n:  # table includes a cls slot
        vit.cls_token.add_(pe[:, 0])
    wb(vit.norm, "norm")
    for i, layer in enumerate(vit.layers):
        p = f"blocks.{i}"
        wb(layer.sa_norm, f"{p}.norm1")
        wb(layer.mlp_norm, f"{p}.norm2")
        qkv_w, qkv_b = w.pop(f"{p}.attn.qkv.weight"), w.pop(f"{p}.attn.qkv.bias")
        for proj, ww, bb in zip((layer.sa.q_proj, layer.sa.k_proj, layer.sa.v_proj), qkv_w.chunk(3, 0), qkv_b.chunk(3, 0)):
            proj.weight.copy_(ww)
            proj.bias.copy_(bb)
        wb(layer.sa.out_proj, f"{p}.attn.proj")
        wb(layer.mlp.linear1, f"{p}.mlp.fc1")
        wb(layer.mlp.linear2, f"{p}.mlp.fc2")
        for target, names in ((layer.sa.out_proj, ("gamma_1", "ls1.gamma")), (layer.mlp.linear2, ("gamma_2", "ls2.gamma"))):
            scale = next((w.pop(f"{p}.{nm}") for nm in names if f"{p}.{nm}" in w), None)
            if scale is not None:
                target.weight.mul_(scale.view(-1, 1))
                target.bias.mul_(scale)
    return sorted(w)


@torch.no_grad()
def load_openai_whisper(model, state_dict: Mapping[str, Tensor]) -> list[str]:
    """OpenAI ``model_state_dict``.  A missing bias (OpenAI's key projections have none) becomes zeros."""
    w = dict(state_dict)

    def wb(mod, key):
        mod.weight.copy_(w.pop(key + ".weight"))
        if getattr(mod, "bias", None) is not None:
            b = w.pop(key + ".bias", None)
            mod.bias.zero_() if b is None else mod.bias.copy_(b)

    enc, dec = model.encoder, model.decoder
    wb(enc.stem[0], "encoder.conv1")
    wb(enc.stem[2], "encoder.conv2")
    enc.pos_embs.copy_(w.pop("encoder.positional_embedding"))
    dec.token_embs.weight.copy_(w.pop("decoder.token_embedding.weight"))
    dec.pos_embs.copy_(w.pop("decoder.positional_embedding"))
    for side, name in ((enc, "encoder"), (dec, "decoder")):
        for i, layer in enumerate(side.layers):
            p = f"{name}.blocks.{i}"
            for mha, ln, key in ((layer.sa, layer.sa_norm, "attn"), (layer.ca, layer.ca_norm, "cross_attn")):
                if mha is None:
                    continue
                wb(mha.q_proj, f"{p}.{key}.query")
                wb(mha.k_proj, f"{p}.{key}.key")
                wb(mha.v_proj, f"{p}.{key}.value")
                wb(mha.out_proj, f"{p}.{key}.out")
                wb(ln, f"{p}.{key}_ln")
            wb(layer.mlp.linear1, f"{p}.mlp.0")
            wb(layer.mlp.linear2, f"{p}.mlp.2")
            wb(layer.mlp_norm, f"{p}.mlp_ln")
        wb(side.norm, "encoder.ln_post" if name == "encoder" else "decoder.ln")
    return sorted(w)
