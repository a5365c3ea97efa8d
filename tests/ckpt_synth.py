"""Synthetic checkpoints in the UPSTREAM formats the reference's converters read (data, built from geometry):

* vision_transformer / big_vision Flax ``.npz`` (reference loader: pytorch_models/image/vit.py:151-200,309-335)
* facebook / timm ViT state_dict with fused qkv and optional layer scale (vit.py:257-306)
* OpenAI Whisper state_dict (pytorch_models/audio2text/whisper.py:96-135)

Values come from synthweights.synth_tensor keyed by the upstream key, so the golden generator (which feeds them to
the reference's converters) and the tests (which feed them to this repo's converters) build identical inputs."""
import numpy as np
import torch

from synthweights import synth_tensor


def _t(key, shape, seed):
    return synth_tensor("ckpt:" + key, shape, seed)


def flax_vit(n_layers, d, h, patch, n_patches, *, big_vision: bool, cls: bool, map_head: bool, seed=0, prefix=""):
    hd = d // h
    if big_vision:
        ln1, mha, ln2, mlp = "LayerNorm_0", "MultiHeadDotProductAttention_0", "LayerNorm_1", "MlpBlock_0"
    else:
        ln1, mha, ln2, mlp = "LayerNorm_0", "MultiHeadDotProductAttention_1", "LayerNorm_2", "MlpBlock_3"
    sd = {}

    def put(k, shape):
        sd[prefix + k] = _t(k, shape, seed).numpy()

    if cls:
        put("cls", (1, 1, d))
    if big_vision:
        put("pos_embedding", (1, n_patches, d))
    else:
        put("Transformer/posembed_input/pos_embedding", (1, n_patches + 1, d))
    put("embedding/kernel", (patch, patch, 3, d))
    put("embedding/bias", (d,))
    put("Transformer/encoder_norm/scale", (d,))
    put("Transformer/encoder_norm/bias", (d,))

    def put_mha(base):
        for nm in ("query", "key", "value"):
            put(f"{base}/{nm}/kernel", (d, h, hd))
            put(f"{base}/{nm}/bias", (h, hd))
        put(f"{base}/out/kernel", (h, hd, d))
        put(f"{base}/out/bias", (d,))

    for i in range(n_layers):
        b = f"Transformer/encoderblock_{i}"
        put(f"{b}/{ln1}/scale", (d,))
        put(f"{b}/{ln1}/bias", (d,))
        put_mha(f"{b}/{mha}")
        put(f"{b}/{ln2}/scale", (d,))
        put(f"{b}/{ln2}/bias", (d,))
        put(f"{b}/{mlp}/Dense_0/kernel", (d, 4 * d))
        put(f"{b}/{mlp}/Dense_0/bias", (4 * d,))
        put(f"{b}/{mlp}/Dense_1/kernel", (4 * d, d))
        put(f"{b}/{mlp}/Dense_1/bias", (d,))
    if map_head:
        put("MAPHead_0/probe", (1, 1, d))
        put_mha("MAPHead_0/MultiHeadDotProductAttention_0")
        put("MAPHead_0/LayerNorm_0/scale", (d,))
        put("MAPHead_0/LayerNorm_0/bias", (d,))
        put("MAPHead_0/MlpBlock_0/Dense_0/kernel", (d, 4 * d))
        put("MAPHead_0/MlpBlock_0/Dense_0/bias", (4 * d,))
        put("MAPHead_0/MlpBlock_0/Dense_1/kernel", (4 * d, d))
        put("MAPHead_0/MlpBlock_0/Dense_1/bias", (d,))
    return sd


def facebook_vit(n_layers, d, patch, n_patches, *, pe_has_cls: bool, layer_scale: str | None, seed=0):
    """layer_scale: None | "gamma" (deit3: gamma_1 / gamma_2) | "ls" (dinov2: ls1.gamma / ls2.gamma)."""
    sd = {}

    def put(k, shape):
        sd[k] = _t(k, shape, seed)

    put("patch_embed.proj.weight", (d, 3, patch, patch))
    put("patch_embed.proj.bias", (d,))
    put("pos_embed", (1, n_patches + int(pe_has_cls), d))
    put("cls_token", (1, 1, d))
    put("norm.weight", (d,))
    put("norm.bias", (d,))
    for i in range(n_layers):
        p = f"blocks.{i}"
        for nm in ("norm1", "norm2"):
            put(f"{p}.{nm}.weight", (d,))
            put(f"{p}.{nm}.bias", (d,))
        put(f"{p}.attn.qkv.weight", (3 * d, d))
        put(f"{p}.attn.qkv.bias", (3 * d,))
        put(f"{p}.attn.proj.weight", (d, d))
        put(f"{p}.attn.proj.bias", (d,))
        put(f"{p}.mlp.fc1.weight", (4 * d, d))
        put(f"{p}.mlp.fc1.bias", (4 * d,))
        put(f"{p}.mlp.fc2.weight", (d, 4 * d))
        put(f"{p}.mlp.fc2.bias", (d,))
        if layer_scale == "gamma":
            put(f"{p}.gamma_1", (d,))
            put(f"{p}.gamma_2", (d,))
        elif layer_scale == "ls":
            put(f"{p}.ls1.gamma", (d,))
            put(f"{p}.ls2.gamma", (d,))
    return sd


def openai_whisper(n_layers, d, n_mels, vocab, seed=0):
    sd = {}

    def put(k, shape):
        sd[k] = _t(k, shape, seed)

    put("encoder.conv1.weight", (d, n_mels, 3))
    put("encoder.conv1.bias", (d,))
    put("encoder.conv2.weight", (d, d, 3))
    put("encoder.conv2.bias", (d,))
    put("encoder.positional_embedding", (1500, d))
    put("decoder.token_embedding.weight", (vocab, d))
    put("decoder.positional_embedding", (448, d))
    for side, cross in (("encoder", False), ("decoder", True)):
        for i in range(n_layers):
            p = f"{side}.blocks.{i}"
            for att in ["attn"] + (["cross_attn"] if cross else []):
                put(f"{p}.{att}.query.weight", (d, d))
                put(f"{p}.{att}.query.bias", (d,))
                put(f"{p}.{att}.key.weight", (d, d))  # OpenAI's key projection has no bias
                put(f"{p}.{att}.value.weight", (d, d))
                put(f"{p}.{att}.value.bias", (d,))
                put(f"{p}.{att}.out.weight", (d, d))
                put(f"{p}.{att}.out.bias", (d,))
                put(f"{p}.{att}_ln.weight", (d,))
                put(f"{p}.{att}_ln.bias", (d,))
            put(f"{p}.mlp.0.weight", (4 * d, d))
            put(f"{p}.mlp.0.bias", (4 * d,))
            put(f"{p}.mlp.2.weight", (d, 4 * d))
            put(f"{p}.mlp.2.bias", (d,))
            put(f"{p}.mlp_ln.weight", (d,))
            put(f"{p}.mlp_ln.bias", (d,))
    put("encoder.ln_post.weight", (d,))
    put("encoder.ln_post.bias", (d,))
    put("decoder.ln.weight", (d,))
    put("decoder.ln.bias", (d,))
    return sd


def hf_gpt2(n_layers, d, vocab, max_pos, seed=0):
    """GPT2LMHeadModel layout: Conv1D weights are (in, out), c_attn packs q|k|v along the OUT axis, keys under transformer."""
    sd = {}

    def put(k, shape):
        sd["transformer." + k] = _t(k, shape, seed)

    put("wte.weight", (vocab, d))
    put("wpe.weight", (max_pos, d))
    for i in range(n_layers):
        b = f"h.{i}."
        for ln in ("ln_1", "ln_2"):
            put(b + ln + ".weight", (d,))
            put(b + ln + ".bias", (d,))
        put(b + "attn.c_attn.weight", (d, 3 * d))
        put(b + "attn.c_attn.bias", (3 * d,))
        put(b + "attn.c_proj.weight", (d, d))
        put(b + "attn.c_proj.bias", (d,))
        put(b + "mlp.c_fc.weight", (d, 4 * d))
        put(b + "mlp.c_fc.bias", (4 * d,))
        put(b + "mlp.c_proj.weight", (4 * d, d))
        put(b + "mlp.c_proj.bias", (d,))
    put("ln_f.weight", (d,))
    put("ln_f.bias", (d,))
    return sd


def hf_bert(n_layers, d, vocab, max_pos, *, roberta: bool, seed=0):
    """BertModel / RobertaModel layout (nn.Linear weights (out, in)); RoBERTa carries two unused leading position rows."""
    sd = {}
    root = "roberta." if roberta else "bert."

    def put(k, shape):
        sd[root + k] = _t(k, shape, seed)

    put("embeddings.word_embeddings.weight", (vocab, d))
    put("embeddings.position_embeddings.weight", (max_pos + (2 if roberta else 0), d))
    put("embeddings.token_type_embeddings.weight", (1 if roberta else 2, d))
    put("embeddings.LayerNorm.weight", (d,))
    put("embeddings.LayerNorm.bias", (d,))
    for i in range(n_layers):
        b = f"encoder.layer.{i}."
        for name, shape in (("attention.self.query", (d, d)), ("attention.self.key", (d, d)), ("attention.self.value", (d, d)),
                            ("attention.output.dense", (d, d)), ("intermediate.dense", (4 * d, d)), ("output.dense", (d, 4 * d))):
            put(b + name + ".weight", shape)
            put(b + name + ".bias", (shape[0],))
        for ln in ("attention.output.LayerNorm", "output.LayerNorm"):
            put(b + ln + ".weight", (d,))
            put(b + ln + ".bias", (d,))
    return sd


def openai_gpt_params(n_layers, d, vocab, max_pos, seed=0):
    """openai/finetune-transformer-lm parameter list after the reference's split / reshape (gpt.py:40-52): positions,
    tokens, then per layer [c_attn w (1, d, 3d), b, c_proj w (1, d, d), b, ln_1 g, b, c_fc w (1, d, 4d), b, c_proj w
    (1, 4d, d), b, ln_2 g, b]."""
    ps = [_t("pos", (max_pos, d), seed), _t("tok", (vocab, d), seed)]
    for i in range(n_layers):
        for j, shape in enumerate(((1, d, 3 * d), (3 * d,), (1, d, d), (d,), (d,), (d,), (1, d, 4 * d), (4 * d,), (1, 4 * d, d), (d,),
                                   (d,), (d,))):
            ps.append(_t(f"l{i}.{j}", shape, seed))
    return ps


def hf_wav2vec2(kind, n_layers, d, dims, kernels, *, legacy: bool, stem_bias: bool, pe_kernel: int, seed=0):
    """Wav2Vec2Model / HubertModel ("wav2vec2"), Data2VecAudioModel ("data2vec") or SEWModel ("sew") layout: conv stem,
    feature projection, weight-normed (g, v) positional conv (five plain convs for data2vec), post-LN-named encoder."""
    sd = {}

    def put(k, shape):
        sd[k] = _t(k, shape, seed)

    c_in = 1
    for i, (c, k) in enumerate(zip(dims, kernels)):
        b = f"feature_extractor.conv_layers.{i}."
        put(b + "conv.weight", (c, c_in, k))
        if stem_bias:
            put(b + "conv.bias", (c,))
        if not legacy or i == 0:
            put(b + "layer_norm.weight", (c,))
            put(b + "layer_norm.bias", (c,))
        c_in = c
    ln, lin = ("layer_norm", "feature_projection") if kind == "sew" else ("feature_projection.layer_norm", "feature_projection.projection")
    put(ln + ".weight", (c_in,))
    put(ln + ".bias", (c_in,))
    if c_in != d:
        put(lin + ".weight", (d, c_in))
        put(lin + ".bias", (d,))
    if kind == "data2vec":
        for i in range(5):
            put(f"encoder.pos_conv_embed.layers.{i}.conv.weight", (d, d // 16, pe_kernel))
            put(f"encoder.pos_conv_embed.layers.{i}.conv.bias", (d,))
    else:
        sd["encoder.pos_conv_embed.conv.weight_g"] = _t("pe.weight_g", (1, 1, pe_kernel), seed).abs() + 0.5
        put("encoder.pos_conv_embed.conv.weight_v", (d, d // 16, pe_kernel))
        put("encoder.pos_conv_embed.conv.bias", (d,))
    put("encoder.layer_norm.weight", (d,))
    put("encoder.layer_norm.bias", (d,))
    for i in range(n_layers):
        b = f"encoder.layers.{i}."
        for name, shape in (("attention.q_proj", (d, d)), ("attention.k_proj", (d, d)), ("attention.v_proj", (d, d)),
                            ("attention.out_proj", (d, d)), ("feed_forward.intermediate_dense", (4 * d, d)),
                            ("feed_forward.output_dense", (d, 4 * d))):
            put(b + name + ".weight", shape)
            put(b + name + ".bias", (shape[0],))
        for nm in ("layer_norm", "final_layer_norm"):
            put(b + nm + ".weight", (d,))
            put(b + nm + ".bias", (d,))
    if kind == "sew":
        put("encoder.upsample.projection.weight", (2 * d, d))
        put("encoder.upsample.projection.bias", (2 * d,))
    return sd


def t5x_flat(n_layers, dim, n_heads, mlp_dim, vocab, seed=0):
    """Flattened t5x ``target`` tree of a T5 v1.1 model (numpy, kernels stored (in, out), as the reference's
    load_t5x_checkpoint returns it - text/t5.py:255-318)."""
    inner = n_heads * 64
    sd = {}

    def put(k, shape):
        sd[k] = _t(k, shape, seed).numpy()

    put("token_embedder.embedding", (vocab, dim))
    put("decoder.logits_dense.kernel", (dim, vocab))
    for side, attns in (("encoder", (("attention", "pre_attention_layer_norm"),)),
                        ("decoder", (("self_attention", "pre_self_attention_layer_norm"),
                                     ("encoder_decoder_attention", "pre_cross_attention_layer_norm")))):
        put(f"{side}.relpos_bias.rel_embedding", (n_heads, 32))
        put(f"{side}.{side}_norm.scale", (dim,))
        for i in range(n_layers):
            b = f"{side}.layers_{i}."
            for attn, norm in attns:
                put(b + norm + ".scale", (dim,))
                for nm in ("query", "key", "value"):
                    put(b + attn + f".{nm}.kernel", (dim, inner))
                put(b + attn + ".out.kernel", (inner, dim))
            put(b + "pre_mlp_layer_norm.scale", (dim,))
            put(b + "mlp.wi_0.kernel", (dim, mlp_dim))
            put(b + "mlp.wi_1.kernel", (dim, mlp_dim))
            put(b + "mlp.wo.kernel", (mlp_dim, dim))
    return sd


def apple_mobilevit(channels, d_models, out_dim, expansion, n_classes=10, seed=0):
    """cvnets v0.1 MobileViT checkpoint (reference loader: pytorch_models/image/mobile_vit.py:125-196): `block.conv` / `block.norm`
    pairs, MobileNetV2 blocks `exp_1x1 / conv_3x3 / red_1x1`, MobileViT blocks `local_rep / global_rep.{i} / conv_proj / fusion`
    with a fused `qkv_proj`, and a classifier the loader drops."""
    sd = {}

    def put(k, shape):
        sd[k] = _t(k, shape, seed)

    def conv_norm(prefix, cin, cout, k, groups=1):
        put(f"{prefix}.block.conv.weight", (cout, cin // groups, k, k))
        put(f"{prefix}.block.norm.weight", (cout,))
        put(f"{prefix}.block.norm.bias", (cout,))
        put(f"{prefix}.block.norm.running_mean", (cout,))
        put(f"{prefix}.block.norm.running_var", (cout,))
        sd[f"{prefix}.block.norm.num_batches_tracked"] = torch.tensor(7)

    def mbconv(prefix, cin, cout):
        hid = cin * expansion
        conv_norm(f"{prefix}.exp_1x1", cin, hid, 1)
        conv_norm(f"{prefix}.conv_3x3", hid, hid, 3, groups=hid)
        conv_norm(f"{prefix}.red_1x1", hid, cout, 1)

    def vit_block(prefix, c, d, n_layers):
        conv_norm(f"{prefix}.local_rep.conv_3x3", c, c, 3)
        put(f"{prefix}.local_rep.conv_1x1.block.conv.weight", (d, c, 1, 1))
        for i in range(n_layers):
            p = f"{prefix}.global_rep.{i}"
            put(f"{p}.pre_norm_mha.0.weight", (d,))
            put(f"{p}.pre_norm_mha.0.bias", (d,))
            put(f"{p}.pre_norm_mha.1.qkv_proj.weight", (3 * d, d))
            put(f"{p}.pre_norm_mha.1.qkv_proj.bias", (3 * d,))
            put(f"{p}.pre_norm_mha.1.out_proj.weight", (d, d))
            put(f"{p}.pre_norm_mha.1.out_proj.bias", (d,))
            put(f"{p}.pre_norm_ffn.0.weight", (d,))
            put(f"{p}.pre_norm_ffn.0.bias", (d,))
            put(f"{p}.pre_norm_ffn.1.weight", (2 * d, d))
            put(f"{p}.pre_norm_ffn.1.bias", (2 * d,))
            put(f"{p}.pre_norm_ffn.4.weight", (d, 2 * d))
            put(f"{p}.pre_norm_ffn.4.bias", (d,))
        put(f"{prefix}.global_rep.{n_layers}.weight", (d,))
        put(f"{prefix}.global_rep.{n_layers}.bias", (d,))
        conv_norm(f"{prefix}.conv_proj", d, c, 1)
        conv_norm(f"{prefix}.fusion", 2 * c, c, 3)

    conv_norm("conv_1", 3, 16, 3)
    mbconv("layer_1.0.block", 16, channels[0])
    mbconv("layer_2.0.block", channels[0], channels[1])
    mbconv("layer_2.1.block", channels[1], channels[1])
    mbconv("layer_2.2.block", channels[1], channels[1])
    for name, cin, c, d, n in (("layer_3", channels[1], channels[2], d_models[0], 2), ("layer_4", channels[2], channels[3], d_models[1], 4),
                               ("layer_5", channels[3], channels[4], d_models[2], 3)):
        mbconv(f"{name}.0.block", cin, c)
        vit_block(f"{name}.1", c, d, n)
    conv_norm("conv_1x1_exp", channels[4], out_dim, 1)
    put("classifier.fc.weight", (n_classes, out_dim))
    put("classifier.fc.bias", (n_classes,))
    return sd


def state_digest(sd) -> dict:
    """name -> [sum, sum |x|, position-weighted sum] (fp64) of every tensor of a loaded model."""
    out = {}
    for k, v in sd.items():
        f = v.detach().double().flatten()
        w = 1.0 + (torch.arange(f.numel(), dtype=torch.float64) % 251) / 251.0
        out[k] = [f.sum().item(), f.abs().sum().item(), (f * w).sum().item()]
    return out
