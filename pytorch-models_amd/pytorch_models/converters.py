"""Weight converters from upstream checkpoint formats into this build's (reference-named) parameters.

Same behaviour as the reference's loaders - vision_transformer / big_vision Flax ``.npz``
(/root/reference pytorch_models/image/vit.py:151-200,309-335), facebook / timm ViT state_dicts with fused qkv and
folded layer scale (vit.py:257-306), OpenAI Whisper state_dicts (audio2text/whisper.py:96-135) - except that
nothing is downloaded: callers pass a local file or an in-memory mapping.  The Hugging Face layouts of the text and audio
models (text/bert.py:79-107, text/gpt2.py:49-81, text/gpt.py:40-84, audio/wav2vec2.py:113-152, audio/data2vec_audio.py:37-71,
audio/sew.py:41-80) are rows of (module path here, upstream name) tables walked by one helper.  Parameters are updated in place with
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


# ---------------------------------------------------------------- table-driven loaders (text / audio, Hugging Face layouts)
def _module_at(root: nn.Module, path: str) -> nn.Module:
    for part in path.split("."):
        root = root[int(part)] if part.isdigit() else getattr(root, part)
    return root


def _take(root: nn.Module, w: dict, rows, at: str = "", transpose: bool = False) -> None:
    """For every (path under root, upstream name) row: weight <- w[at + name + ".weight"] (transposed for Conv1D-style (in, out)
    matrices), bias likewise when the module has one.  Consumed keys leave ``w``."""
    for path, name in rows:
        mod = _module_at(root, path)
        val = w.pop(f"{at}{name}.weight")
        mod.weight.copy_(val.T if transpose and val.ndim == 2 else val)
        if getattr(mod, "bias", None) is not None:
            mod.bias.copy_(w.pop(f"{at}{name}.bias"))


def _split3(mha, weight: Tensor, bias: Tensor) -> None:
    """A fused (d, 3 d) Conv1D projection -> q / k / v nn.Linear."""
    for proj, ww, bb in zip((mha.q_proj, mha.k_proj, mha.v_proj), weight.chunk(3, -1), bias.chunk(3, -1)):
        proj.weight.copy_(ww.T)
        proj.bias.copy_(bb)


_BERT_LAYER = (("sa.q_proj", "attention.self.query"), ("sa.k_proj", "attention.self.key"), ("sa.v_proj", "attention.self.value"),
               ("sa.out_proj", "attention.output.dense"), ("sa_norm", "attention.output.LayerNorm"),
               ("mlp.linear1", "intermediate.dense"), ("mlp.linear2", "output.dense"), ("mlp_norm", "output.LayerNorm"))
_GPT2_LAYER = (("sa_norm", "ln_1"), ("sa.out_proj", "attn.c_proj"), ("mlp_norm", "ln_2"), ("mlp.linear1", "mlp.c_fc"),
               ("mlp.linear2", "mlp.c_proj"))
_W2V_LAYER = (("sa.q_proj", "attention.q_proj"), ("sa.k_proj", "attention.k_proj"), ("sa.v_proj", "attention.v_proj"),
              ("sa.out_proj", "attention.out_proj"), ("sa_norm", "layer_norm"), ("mlp.linear1", "feed_forward.intermediate_dense"),
              ("mlp.linear2", "feed_forward.output_dense"), ("mlp_norm", "final_layer_norm"))


@torch.no_grad()
def load_hf_bert(model, state_dict: Mapping[str, Tensor]) -> list[str]:
    """BertModel / RobertaModel.  The vocabulary may be shorter than the padded table; RoBERTa's two unused leading position
    rows are dropped; token-type row 0 is folded into the positions (the class has no token types)."""
    roberta = any(k.startswith("roberta.") for k in state_dict)
    w = {k.removeprefix("bert.").removeprefix("roberta."): v for k, v in state_dict.items()}
    words = w.pop("embeddings.word_embeddings.weight")
    model.token_embs.weight[: words.shape[0]] = words
    positions = w.pop("embeddings.position_embeddings.weight")[2 if roberta else 0:]
    model.pos_embs.copy_(positions + w.pop("embeddings.token_type_embeddings.weight")[0])
    _take(model, w, (("norm", "embeddings.LayerNorm"),))
    for i, layer in enumerate(model.layers):
        _take(layer, w, _BERT_LAYER, f"encoder.layer.{i}.")
    return sorted(w)


@torch.no_grad()
def load_hf_gpt2(model, state_dict: Mapping[str, Tensor]) -> list[str]:
    """GPT2LMHeadModel: Conv1D matrices are (in, out) and c_attn holds q, k, v side by side."""
    w = {k.removeprefix("transformer."): v for k, v in state_dict.items()}
    words = w.pop("wte.weight")
    model.token_embs.weight[: words.shape[0]] = words
    model.pos_embs.copy_(w.pop("wpe.weight"))
    for i, layer in enumerate(model.layers):
        at = f"h.{i}."
        _split3(layer.sa, w.pop(at + "attn.c_attn.weight"), w.pop(at + "attn.c_attn.bias"))
        _take(layer, w, _GPT2_LAYER, at, transpose=True)
    _take(model, w, (("norm", "ln_f"),))
    return sorted(w)


@torch.no_grad()
def load_openai_gpt(model, params) -> None:
    """openai/finetune-transformer-lm: positions, tokens, then per layer [c_attn w, b, c_proj w, b, ln_1 g, b, c_fc w, b,
    c_proj w, b, ln_2 g, b], matrices stored (1, in, out)."""
    it = iter(torch.as_tensor(p) for p in params)
    model.pos_embs.copy_(next(it))
    words = next(it)
    model.token_embs.weight[: words.shape[0]] = words
    for layer in model.layers:
        _split3(layer.sa, next(it).squeeze(0), next(it))
        for mod, is_matrix in ((layer.sa.out_proj, True), (layer.sa_norm, False), (layer.mlp.linear1, True), (layer.mlp.linear2, True),
                               (layer.mlp_norm, False)):
            weight = next(it)
            mod.weight.copy_(weight.squeeze(0).T if is_matrix else weight)
            mod.bias.copy_(next(it))


@torch.no_grad()
def load_hf_wav2vec2(model, state_dict: Mapping[str, Tensor], *, flavour: str = "wav2vec2") -> list[str]:
    """Wav2Vec2Model / HubertModel ("wav2vec2"), Data2VecAudioModel ("data2vec"), SEWModel ("sew"): conv stem, feature
    projection, positional conv (weight-normed: w = g v / ||v|| per tap, undone here; five plain convs for data2vec), encoder."""
    w = dict(state_dict)
    for i, blk in enumerate(model.feature_encoder):
        rows = [("0", "conv")] + ([] if isinstance(blk[2], nn.Identity) else [("2", "layer_norm")])
        _take(blk, w, rows, f"feature_extractor.conv_layers.{i}.")
    ln, lin = ("layer_norm", "feature_projection") if flavour == "sew" else ("feature_projection.layer_norm", "feature_projection.projection")
    _take(model.proj, w, [("0", ln)] + ([("1", lin)] if len(model.proj) > 1 else []))
    if flavour == "data2vec":
        _take(model.pe_conv, w, [(f"{i}.0", f"encoder.pos_conv_embed.layers.{i}.conv") for i in range(len(model.pe_conv))])
    else:
        g, v = w.pop("encoder.pos_conv_embed.conv.weight_g"), w.pop("encoder.pos_conv_embed.conv.weight_v")
        norm = v.float().pow(2).sum((0, 1), keepdim=True).sqrt().clamp_min(1e-12).to(v.dtype)  # weight_norm(dim=2): per tap
        model.pe_conv[1].weight.copy_(g * v / norm)
        model.pe_conv[1].bias.copy_(w.pop("encoder.pos_conv_embed.conv.bias"))
    _take(model, w, (("norm", "encoder.layer_norm"),))
    for i, layer in enumerate(model.layers):
        _take(layer, w, _W2V_LAYER, f"encoder.layers.{i}.")
    if flavour == "sew":
        _take(model.upsample, w, (("0", "encoder.upsample.projection"),))
    return sorted(w)
