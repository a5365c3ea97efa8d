"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE on CPU (build container only).

    python tests/golden/make_golden.py            # needs /root/reference; never runs on the GPU box

Weights and inputs come from ``synthweights`` (numpy PCG64 keyed by parameter name), so the
fixtures hold only expected OUTPUTS (or slices / digests of large ones) plus the seeds and
shapes needed to regenerate the inputs.  Nothing of the reference's source travels: a fixture is
data.  The tests in tests/test_oracle_golden.py pin the oracle (oracle/) to these vectors.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")  # the reference's ``pytorch_models`` wins
sys.path.insert(1, os.path.join(ROOT, "pytorch-models_amd"))  # only for ``synthweights``

import pytorch_models  # noqa: E402

assert pytorch_models.__file__.startswith("/root/reference"), pytorch_models.__file__
from pytorch_models.audio.spectrogram import MelSpectrogram, Spectrogram, get_mel_filters  # noqa: E402
from pytorch_models.audio2text import Whisper, WhisperDecoder, WhisperEncoder, WhisperPreprocessor  # noqa: E402
from pytorch_models.image import ViT  # noqa: E402
from pytorch_models.transformer import MHA, Decoder, DecoderLayer, Encoder, EncoderLayer  # noqa: E402
from synthweights import fill_module, synth_input, synth_tokens  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def save(name, meta=None, **arrays):
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    out["meta"] = np.frombuffer(json.dumps(meta or {}).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def digest(t: torch.Tensor) -> np.ndarray:
    """Order-sensitive digest of a large tensor: [sum, sum|x|, sum x*w] in fp64, w = 1 + (i mod 251)/251."""
    f = t.double().flatten()
    w = 1.0 + (torch.arange(f.numel(), dtype=torch.float64) % 251) / 251.0
    return np.array([f.sum().item(), f.abs().sum().item(), (f * w).sum().item()])


# ---------------------------------------------------------------- 1/2: transformer blocks
def g_blocks():
    d = 64
    out = {}
    x = synth_input("blk_x", (2, 10, d), 1)
    mem = synth_input("blk_mem", (2, 7, d), 1)
    for pre in (True, False):
        for eps in (1e-5, 1e-6):
            m = EncoderLayer(d, pre_norm=pre, norm_eps=eps).eval()
            fill_module(m, 11)
            out[f"enc_pre{int(pre)}_eps{eps}"] = m(x)
            m = DecoderLayer(d, cross_attn=True, pre_norm=pre, norm_eps=eps).eval()
            fill_module(m, 12)
            out[f"dec_pre{int(pre)}_eps{eps}"] = m(x, mem)
    m = DecoderLayer(d, cross_attn=False).eval()  # decoder-only (GPT-style) path of transformer.py:99
    fill_module(m, 13)
    out["dec_nocross"] = m(x)
    for act in ("gelu", "approximate_gelu", "relu", "silu"):
        m = EncoderLayer(d, act=act).eval()
        fill_module(m, 14)
        out[f"enc_act_{act}"] = m(x)
    m = Encoder(3, 128, n_heads=2).eval()
    fill_module(m, 15)
    out["encoder3"] = m(synth_input("blk_x128", (2, 9, 128), 1))
    m = Decoder(2, 128, cross_attn=True).eval()
    fill_module(m, 16)
    out["decoder2"] = m(synth_input("blk_x128", (2, 9, 128), 1), synth_input("blk_mem128", (2, 5, 128), 1))
    save("blocks", dict(d=d), **out)


def g_mha():
    d = 64
    q = synth_input("mha_q", (2, 6, d), 2)
    k = synth_input("mha_k", (2, 9, d), 2)
    v = synth_input("mha_v", (2, 9, d), 2)
    bias = synth_input("mha_bias", (2, 1, 6, 9), 2)
    out = {}
    m = MHA(d).eval()  # default: head_dim 64 -> 1 head
    fill_module(m, 21)
    out["default_q"] = m(q)
    m = MHA(d, n_heads=4).eval()
    fill_module(m, 22)
    out["h4_q"] = m(q)
    out["h4_qk"] = m(q, k)
    out["h4_qkv"] = m(q, k, v)
    out["h4_bias"] = m(q, k, v, attn_bias=bias)
    # key-padding masks broadcast over heads AND queries: boolean keep-mask and its additive form
    keep = torch.ones(2, 1, 1, 9, dtype=torch.bool)
    keep[0, ..., 6:] = False
    keep[1, ..., 8:] = False
    out["h4_keypad_bool"] = m(q, k, v, attn_bias=keep)
    out["h4_keypad_add"] = m(q, k, v, attn_bias=torch.zeros(2, 1, 1, 9).masked_fill(~keep, float("-inf")))
    out["h4_causal"] = m(q, causal=True)
    out["h4_causal_rect"] = m(q, k, causal=True)  # top-left aligned, L_q != S_k
    out["h4_unbatched"] = m(q[0])  # (L, d) leading-dim-free input (tests/text/test_t5.py:27-29)
    m = MHA(d, n_heads=2, head_dim=16).eval()  # n_heads * head_dim < d_model (transformer.py:18)
    fill_module(m, 23)
    out["h2hd16_q"] = m(q)
    m = MHA(d, head_dim=32, bias=False).eval()
    fill_module(m, 24)
    out["hd32_nobias"] = m(q, k)
    save("mha", dict(d=d), **out)


def g_sdpa_alignment():
    # F3: L_q = 1, S_k = 5, is_causal=True -> only key 0 visible
    q = synth_input("f3_q", (1, 1, 1, 8), 3)
    k = synth_input("f3_k", (1, 1, 5, 8), 3)
    v = synth_input("f3_v", (1, 1, 5, 8), 3)
    out = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)
    save("sdpa_alignment", {}, out=out, v0=v[:, :, 0])


# ---------------------------------------------------------------- ViT
def g_vit():
    out = {}
    m = ViT.from_google("Ti/16").eval()
    fill_module(m, 31)
    x = synth_input("vit_ti", (1, 3, 224, 224), 31)
    out["ti16_b1"] = m(x)
    # intermediate: tokens after patch-embed + pe + cls (vit.py:78-81)
    t = m.patch_embed(x).flatten(-2).transpose(-1, -2) + m.pe
    t = torch.cat([m.cls_token, t], dim=-2)
    out["ti16_tokens_slice"] = t[0, :5, :16]
    out["ti16_tokens_digest"] = digest(t)
    # resize_pe(256) then forward at 256 (tests/image/test_vit.py:21-26)
    m.resize_pe(256)
    out["ti16_pe256"] = m.pe.detach().clone()
    out["ti16_b1_256"] = m(synth_input("vit_ti256", (1, 3, 256, 256), 31))

    m = ViT.from_google("B/16").eval()
    fill_module(m, 32)
    xb = synth_input("vit_b", (4, 3, 224, 224), 32)
    out["b16_first4"] = torch.cat([m(xb[i : i + 1]) for i in range(4)])  # F1: per-sample loop

    m = ViT.from_google("B/16_siglip").eval()
    fill_module(m, 33)
    out["b16_siglip_b2"] = m(synth_input("vit_bs", (2, 3, 224, 224), 33))

    m = ViT.from_google("L/16_siglip", img_size=384).eval()
    fill_module(m, 34)
    out["l16_siglip384_b2"] = m(synth_input("vit_ls", (2, 3, 384, 384), 34))

    m = ViT.from_facebook("S/14_dinov2").eval()  # img 518, patch 14 -> L = 1370
    fill_module(m, 35)
    out["s14_dinov2_b1"] = m(synth_input("vit_dv2", (1, 3, 518, 518), 35))

    m = ViT(2, 128, 2, 16, img_size=64, pool_type="gap").eval()
    fill_module(m, 36)
    out["tiny_gap_b1"] = m(synth_input("vit_gap", (1, 3, 64, 64), 36))
    m = ViT(2, 128, 2, 16, img_size=64, cls_token=False, pool_type="gap").eval()
    fill_module(m, 37)
    out["tiny_gap_nocls_b3"] = m(synth_input("vit_gap3", (3, 3, 64, 64), 37))
    save("vit", {}, **out)


# ---------------------------------------------------------------- spectrogram / log-mel
def g_audio():
    out = {}
    out["filters80"] = get_mel_filters(80, 400, 16000)
    out["filters128"] = get_mel_filters(128, 400, 16000)
    x1 = synth_input("wave_1s", (16000,), 41)
    out["spec_1s"] = Spectrogram(400, 160)(x1)  # (201, 101)
    out["mel_1s"] = MelSpectrogram(400, 160, 80, 16000)(x1)  # (80, 101)
    out["logmel_1s"] = WhisperPreprocessor()(x1)  # (80, 100)
    out["logmel128_1s"] = WhisperPreprocessor("large-v3")(x1)
    x30 = synth_input("wave_30s", (2, 480000), 42, scale=0.1)
    lm = WhisperPreprocessor("base")(x30)  # (2, 80, 3000)
    out["logmel_30s_digest"] = digest(lm)
    out["logmel_30s_slice"] = lm[:, ::8, ::100]
    out["logmel_30s_head"] = lm[0, :, :8]
    out["logmel_30s_tail"] = lm[1, :, -8:]
    # tests/audio2text/test_whisper.py:57-65: batched == per-sample (F4) with a DC offset on sample 0
    xb = synth_input("wave_batch", (4, 16000), 43)
    xb[0] += 0.5
    out["logmel_batch"] = WhisperPreprocessor()(xb)
    # a silent clip: clamp(0).log10() = -inf everywhere, max - 8 = -inf -> stays -inf (edge case of whisper.py:145-146)
    xs = synth_input("wave_half_silent", (16000,), 44)
    xs[8000:] = 0
    out["logmel_half_silent"] = WhisperPreprocessor()(xs)
    save("audio", {}, **out)


# ---------------------------------------------------------------- Whisper
def g_whisper():
    out = {}
    # smoke shapes of tests/audio2text/test_whisper.py:10-17
    vocab, L, d = 100, 2, 64
    enc = WhisperEncoder(L, d).eval()
    fill_module(enc, 51)
    mel = synth_input("w_mel16", (2, 80, 16), 51)
    out["smoke_encoder"] = enc(mel)
    dec = WhisperDecoder(vocab, L, d).eval()
    fill_module(dec, 52)
    toks = synth_tokens("w_tok32", (2, 32), vocab, 52)
    out["smoke_decoder"] = dec(toks, synth_input("w_mem16", (2, 16, d), 52))
    w = Whisper(vocab, L, d).eval()
    fill_module(w, 53)
    out["smoke_whisper"] = w(mel, toks)

    # reference-test shape (tests/audio2text/test_whisper.py:39-42) on "tiny" geometry
    w = Whisper.from_openai("tiny").eval()
    fill_module(w, 54)
    mel = synth_input("w_mel3000", (1, 80, 3000), 54)
    toks = synth_tokens("w_tok200", (1, 32), 200, 54)
    memory = w.encoder(mel)
    logits = w.decoder(toks, memory)
    out["tiny_memory_digest"] = digest(memory)
    out["tiny_memory_slice"] = memory[0, ::100, ::32]
    out["tiny_logits_digest"] = digest(logits)
    out["tiny_logits_slice"] = logits[0, :, :128]
    out["tiny_logits_argmax"] = logits.argmax(-1)
    out["tiny_logits_max"] = logits.max(-1).values

    # greedy ids by full-prefix recompute: B=2, prompt 4, 32 new tokens, with top1-top2 margins
    for tag, seed in (("tiny", 55), ("base", 56)):
        w = Whisper.from_openai(tag).eval()
        fill_module(w, seed)
        wave = synth_input(f"w_wave_{tag}", (2, 480000), seed, scale=0.1)
        mel = WhisperPreprocessor(tag)(wave)
        memory = w.encoder(mel)
        toks = synth_tokens(f"w_prompt_{tag}", (2, 4), 51865, seed)
        margins = []
        for _ in range(32):
            last = w.decoder(toks, memory)[:, -1]
            top2 = last.topk(2, -1)
            margins.append(top2.values[:, 0] - top2.values[:, 1])
            toks = torch.cat([toks, top2.indices[:, :1]], 1)
        out[f"greedy_{tag}_tokens"] = toks
        out[f"greedy_{tag}_margins"] = torch.stack(margins, 1)
        out[f"greedy_{tag}_memory_digest"] = digest(memory)
        out[f"greedy_{tag}_memory_slice"] = memory[:, ::100, ::32]
        print(tag, "layers", len(w.encoder.layers), "min margin", float(torch.stack(margins, 1).min()))

    # the full-length run of BASELINE configs[2]'s decode (prompt 4, 224 new tokens), B = 2, by the reference's full-prefix
    # loop: once on the reference's plain fp32 weights ("f"), once on bf16-representable weights ("r": what a bf16 model on
    # the GPU holds, so that the fp32 reference forward and the exact mode of the HIP path see the SAME weights)
    from synthweights import bf16_round_

    for tag, seed in (("tiny", 55), ("base", 56)):
        for kind in ("f", "r"):
            w = Whisper.from_openai(tag).eval()
            fill_module(w, seed)
            if kind == "r":
                bf16_round_(w)
            wave = synth_input(f"w_wave_{tag}", (2, 480000), seed, scale=0.1)
            memory = w.encoder(WhisperPreprocessor(tag)(wave))
            toks = synth_tokens(f"w_prompt_{tag}", (2, 4), 51865, seed)
            margins = []
            for _ in range(224):
                last = w.decoder(toks, memory)[:, -1]
                top2 = last.topk(2, -1)
                margins.append(top2.values[:, 0] - top2.values[:, 1])
                toks = torch.cat([toks, top2.indices[:, :1]], 1)
            out[f"greedy224{kind}_{tag}_tokens"] = toks
            out[f"greedy224{kind}_{tag}_margins"] = torch.stack(margins, 1)
            out[f"greedy224{kind}_{tag}_memory_digest"] = digest(memory)
            print(tag, kind, "224 tokens: min margin", float(torch.stack(margins, 1).min()))
    save("whisper", {}, **out)


def g_geometry():
    """Constructor contract: parameter names / shapes / counts of every hot-path constructor."""
    rec = {}
    for tag in ("Ti/16", "B/16", "B/16_siglip"):
        m = ViT.from_google(tag)
        rec["google:" + tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
    m = ViT.from_google("L/16_siglip", img_size=384)
    rec["google:L/16_siglip@384"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    for tag in ("S/16_deit3", "S/16_dino", "S/14_dinov2"):
        m = ViT.from_facebook(tag)
        rec["facebook:" + tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
    for tag in ("tiny", "tiny.en", "base", "large-v3"):
        with torch.device("meta"):
            m = Whisper.from_openai(tag)
        rec["openai:" + tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
    p = WhisperPreprocessor()
    rec["preprocessor"] = {k: list(v.shape) for k, v in p.state_dict().items()}
    with open(os.path.join(HERE, "geometry.json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)
    print("geometry.json", os.path.getsize(os.path.join(HERE, "geometry.json")) // 1024, "KiB")


def g_converters():
    """Weight converters: feed synthetic upstream-format checkpoints (tests/ckpt_synth.py) to the
    reference's loaders and record a digest of every resulting parameter."""
    import tempfile

    import pytorch_models.image.vit as ref_vit_mod

    sys.path.insert(2, os.path.join(ROOT, "tests"))
    import ckpt_synth as C

    rec = {}
    with tempfile.TemporaryDirectory() as tmp:
        def load_flax(m, ck, **kw):
            path = os.path.join(tmp, "ck.npz")
            np.savez(path, **ck)
            ref_vit_mod.torch_hub_download = lambda url, *a, **k: path  # no network: serve the local synthetic file
            m.load_flax_ckpt("synthetic.npz", **kw)

        m = ViT(2, 128, 2, 16, img_size=64)
        load_flax(m, C.flax_vit(2, 128, 2, 16, 16, big_vision=False, cls=True, map_head=False, seed=61))
        rec["flax_augreg"] = C.state_digest(m.state_dict())
        m = ViT(2, 128, 2, 16, img_size=64, cls_token=False, pool_type="mha")
        load_flax(m, C.flax_vit(2, 128, 2, 16, 16, big_vision=True, cls=False, map_head=True, seed=62, prefix="params/img/"),
                  big_vision=True, prefix="params/img/")
        rec["flax_siglip"] = C.state_digest(m.state_dict())
    m = ViT(2, 128, 2, 16, img_size=64)
    m.load_facebook_state_dict(C.facebook_vit(2, 128, 16, 16, pe_has_cls=False, layer_scale="gamma", seed=63))
    rec["fb_deit3"] = C.state_digest(m.state_dict())
    m = ViT(2, 128, 2, 16, img_size=64)
    m.load_facebook_state_dict(C.facebook_vit(2, 128, 16, 16, pe_has_cls=True, layer_scale="ls", seed=64))
    rec["fb_dinov2"] = C.state_digest(m.state_dict())
    m = ViT(2, 128, 2, 16, img_size=64)
    m.load_facebook_state_dict(C.facebook_vit(2, 128, 16, 16, pe_has_cls=True, layer_scale=None, seed=65))
    rec["fb_dino"] = C.state_digest(m.state_dict())
    w = Whisper(100, 2, 64)
    w.load_openai_state_dict(C.openai_whisper(2, 64, 80, 100, seed=66))
    rec["openai_whisper"] = C.state_digest(w.state_dict())
    with open(os.path.join(HERE, "converters.json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)
    print("converters.json", os.path.getsize(os.path.join(HERE, "converters.json")) // 1024, "KiB")


def g_text():
    """GPT-2 / GPT / BERT forwards on synthweights at the sizes of the reference's own tests
    (tests/text/test_gpt2.py:16, test_gpt.py:16, test_bert.py:16), the greedy loop of text/generator.py with a stub
    tokenizer, and digests of what the reference's HF / OpenAI loaders make of synthetic upstream checkpoints."""
    from pytorch_models.text import BERT, GPT, GPT2, DecoderGenerator

    sys.path.insert(2, os.path.join(ROOT, "tests"))
    import ckpt_synth as C

    out, rec = {}, {}
    tok = synth_tokens("text_tok", (2, 16), 2000, 71)
    m = GPT2(2, 128).eval()
    fill_module(m, 72)
    lg = m(tok)  # (2, 16, 50257): keep every 101st vocabulary column, the argmax and an order-sensitive digest
    out["gpt2_logits_s101"], out["gpt2_argmax"], out["gpt2_digest"] = lg[..., ::101], lg.argmax(-1), digest(lg)

    class Tok:  # the reference's generator wants a tokenizer: ids <-> space-separated decimal strings
        eos_token_id = -1

        def encode(self, s):
            return [int(t) for t in s.split()]

        def decode(self, ids):
            return " ".join(str(int(i)) for i in ids)

    gen = DecoderGenerator(m, Tok())
    out["gpt2_greedy"] = np.array([[int(t) for t in gen.generate(Tok().decode(tok[b, :6]), max_tokens=12, topk=1).split()] for b in range(2)])
    m = GPT(2, 128).eval()
    fill_module(m, 73)
    lg = m(tok)
    out["gpt_logits_s101"], out["gpt_argmax"], out["gpt_digest"] = lg[..., ::101], lg.argmax(-1), digest(lg)
    m = BERT(2000, 2, 128).eval()
    fill_module(m, 74)
    out["bert_hidden"] = m(tok)
    # loaders
    m = GPT2(2, 128)
    m.load_hf_state_dict(C.hf_gpt2(2, 128, GPT2.vocab_size, GPT2.max_seq_len, seed=75))
    rec["hf_gpt2"] = C.state_digest(m.state_dict())
    import io
    from contextlib import redirect_stdout
    for name, rob in (("hf_bert", False), ("hf_roberta", True)):
        m = BERT(1024, 2, 128, max_seq_len=64)
        with redirect_stdout(io.StringIO()):
            m.load_hf_state_dict(C.hf_bert(2, 128, 1024, 64, roberta=rob, seed=76))
        rec[name] = C.state_digest(m.state_dict())
    save("text", dict(tok_seed=71), **out)
    with open(os.path.join(HERE, "text_converters.json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)
    print("text_converters.json", os.path.getsize(os.path.join(HERE, "text_converters.json")) // 1024, "KiB")


def g_audio_enc():
    """the wav2vec2 family on synthweights at the sizes of the reference's own tests
    (tests/audio/test_wav2vec2.py:12,17: x (2, 6400), Wav2Vec2(2, 64)) plus d = 128 variants covering the legacy
    (instance-norm, bias-free) stem, post-norm, data2vec's five-layer positional conv and SEW's down / up-sampling
    (odd frame count), and digests of what the reference's HF loaders make of synthetic upstream checkpoints."""
    import io
    from contextlib import redirect_stdout

    from pytorch_models.audio import SEW, Data2VecAudio, Wav2Vec2

    sys.path.insert(2, os.path.join(ROOT, "tests"))
    import ckpt_synth as C

    out, rec = {}, {}
    x = synth_input("w2v_x", (2, 6400), 81)
    cases = dict(
        w2v_d64=(lambda: Wav2Vec2(2, 64), 82),
        w2v_legacy_post_d128=(lambda: Wav2Vec2(2, 128, stem_bias=False, stem_legacy=True, pre_norm=False), 83),
        d2v_d128=(lambda: Data2VecAudio(2, 128), 84),
        sew_d128=(lambda: SEW(2, 128), 85),
    )
    for name, (make, seed) in cases.items():
        m = make().eval()
        fill_module(m, seed)
        feat = m.feature_encoder(x.unsqueeze(1)).transpose(1, 2)  # (B, T, C)
        out[name + "_feat_s8"] = feat[..., ::8]
        out[name + "_proj"] = m.proj(feat)
        out[name] = m(x)
    # 6400 samples give 19 frames: odd, so SEW appends a zero frame after up-sampling (sew.py:37-38); 6080 give 18
    assert out["sew_d128"].shape[1] % 2 == 1
    m = SEW(2, 128).eval()
    fill_module(m, 85)
    out["sew_d128_even"] = m(x[:, :6080])
    assert out["sew_d128_even"].shape[1] % 2 == 0
    loaders = dict(
        hf_wav2vec2=(lambda: Wav2Vec2(2, 128), dict(kind="wav2vec2", legacy=False, stem_bias=True, pe_kernel=128)),
        hf_wav2vec2_base=(lambda: Wav2Vec2(2, 128, stem_bias=False, stem_legacy=True, pre_norm=False),
                          dict(kind="wav2vec2", legacy=True, stem_bias=False, pe_kernel=128)),
        hf_data2vec=(lambda: Data2VecAudio(2, 128), dict(kind="data2vec", legacy=False, stem_bias=False, pe_kernel=19)),
        hf_sew=(lambda: SEW(2, 128), dict(kind="sew", legacy=True, stem_bias=True, pe_kernel=31)),
    )
    for name, (make, kw) in loaders.items():
        m = make()
        kind = kw.pop("kind")
        with redirect_stdout(io.StringIO()) as so:
            m.load_hf_state_dict(C.hf_wav2vec2(kind, 2, 128, m.STEM_DIMS, m.STEM_KERNELS, seed=86, **kw))
        assert "dict_keys([])" in so.getvalue(), so.getvalue()
        rec[name] = C.state_digest(m.state_dict())
    save("audio_enc", dict(x_seed=81), **out)
    with open(os.path.join(HERE, "audio_converters.json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)
    print("audio_converters.json", os.path.getsize(os.path.join(HERE, "audio_converters.json")) // 1024, "KiB")


def g_t5():
    """T5 v1.1 blocks at the shapes of the reference's own tests (tests/text/test_t5.py:10-16: dim 512, 6 heads of 64 -
    inner width 384 != dim -, mlp 1024; 2 layers here), batched and unbatched, the greedy loop of T5Generator.generate
    on token ids, and a digest of what the reference's inline t5x conversion (t5.py:169-178) makes of a synthetic
    flattened t5x checkpoint."""
    from pytorch_models.text import T5Decoder, T5Encoder, T5Model
    from pytorch_models.text.t5 import _rename_key

    sys.path.insert(2, os.path.join(ROOT, "tests"))
    import ckpt_synth as C

    dim, n_heads, n_layers, mlp_dim = 512, 6, 2, 1024
    out = {}
    x = synth_input("t5_x", (2, 64, dim), 91)
    mem = synth_input("t5_mem", (2, 32, dim), 91)
    m = T5Encoder(dim, n_heads, n_layers, mlp_dim).eval()
    fill_module(m, 92)
    out["encoder"] = m(x)[..., ::4]
    out["encoder_unbatched"] = m(x[0])[..., ::4]
    m = T5Decoder(dim, n_heads, n_layers, mlp_dim).eval()
    fill_module(m, 93)
    out["decoder"] = m(x, mem)[..., ::4]
    m = T5Model(2000, dim, n_heads, n_layers, mlp_dim).eval()
    fill_module(m, 94)
    tok = synth_tokens("t5_tok", (2, 64), 1000, 95)
    tgt = synth_tokens("t5_tgt", (2, 32), 1000, 95)
    lg = m(tok, tgt)
    out["model_logits_s7"], out["model_argmax"], out["model_digest"] = lg[..., ::7], lg.argmax(-1), digest(lg)
    # T5Generator.generate's loop (t5.py:213-227) on ids: pad id 0 first, arg-max of the last position, stop at eos id 1
    ids = [0]
    memory = m.encode(tok[0])
    while len(ids) < 12:
        ids.append(m.decode(torch.tensor(ids), memory).argmax(-1)[-1].item())
        if ids[-1] == 1:
            break
    out["greedy"] = np.array(ids)
    rp = m.encoder.attn_bias
    rp.bias.copy_(torch.arange(32, dtype=torch.float32).expand(n_heads, 32))  # bias value == bucket id
    for L in (8, 64, 200):
        out[f"buckets_bi_{L}"] = rp(L, True)[0].to(torch.int8)
        out[f"buckets_uni_{L}"] = rp(L, False)[0].to(torch.int8)
    # the reference converts a t5x checkpoint inline in from_t5x (needs the network); the same statements on a synthetic one
    ckpt = C.t5x_flat(2, 128, 2, 256, 500, seed=96)
    sd = {}
    for k, v in ckpt.items():
        v = torch.from_numpy(v)
        if k.endswith("kernel"):
            v = v.T
        if k.endswith(("query.kernel", "key.kernel")):
            v = v * 64**0.25
        sd[_rename_key(k)] = v
    m = T5Model(500, 128, 2, 2, 256)
    m.load_state_dict(sd)
    with open(os.path.join(HERE, "t5_converter.json"), "w") as f:
        json.dump(C.state_digest(m.state_dict()), f, indent=0, sort_keys=True)
    save("t5", dict(dim=dim), **out)


def g_mobile_vit():
    """MobileViT xxs / xs / s (image/mobile_vit.py) on 64 x 64 images - the smallest side its five stride-2 stages and 2 x 2
    patches allow -: the outputs of the five stages and the pooled features, every BatchNorm with non-trivial running
    statistics; and a digest of what load_apple_state_dict makes of a synthetic cvnets checkpoint."""
    from pytorch_models.image.mobile_vit import MobileViT

    sys.path.insert(2, os.path.join(ROOT, "tests"))
    import ckpt_synth as C

    out, conv = {}, {}
    x = synth_input("mv_x", (2, 3, 64, 64), 61)
    for v in ("xxs", "xs", "s"):
        m = MobileViT.from_apple(v).eval()
        fill_module(m, 62)
        h = x
        for i in range(5):
            h = m[i](h)
            out[f"{v}_stage{i}"] = h if i < 2 else h  # small at this image size: kept whole
        out[f"{v}_out"] = m[5](h)
        assert torch.allclose(out[f"{v}_out"], m(x))
        channels, d_models, out_dim, expansion = dict(
            xxs=([16, 24, 48, 64, 80], [64, 80, 96], 320, 2), xs=([32, 48, 64, 80, 96], [96, 120, 144], 384, 4),
            s=([32, 64, 96, 128, 160], [144, 192, 240], 640, 4))[v]
        m2 = MobileViT.from_apple(v)
        m2.load_apple_state_dict(C.apple_mobilevit(channels, d_models, out_dim, expansion, seed=63))
        conv[v] = C.state_digest(m2.state_dict())
    with open(os.path.join(HERE, "mobile_vit_converter.json"), "w") as f:
        json.dump(conv, f, indent=0, sort_keys=True)
    save("mobile_vit", dict(img=64), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["blocks", "mha", "sdpa", "vit", "audio", "whisper", "geometry", "converters", "text", "audio_enc", "t5", "mobile_vit"]
    table = dict(blocks=g_blocks, mha=g_mha, sdpa=g_sdpa_alignment, vit=g_vit, audio=g_audio, whisper=g_whisper,
                 geometry=g_geometry, converters=g_converters, text=g_text, audio_enc=g_audio_enc, t5=g_t5, mobile_vit=g_mobile_vit)
    for w in which:
        table[w]()
