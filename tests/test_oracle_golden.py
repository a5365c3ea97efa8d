"""Pin the oracle (oracle/) to golden vectors captured from the reference (tests/golden/make_golden.py).

CPU only.  Weights are regenerated with synthweights from the product modules' parameter names - the
shared contract - so these tests also prove that the product constructors reproduce the reference's
state_dict layout (names + shapes: see also test_constructors.py).
"""
import pytest
import torch

from oracle import ref_spectrogram as RS
from oracle import ref_transformer as RT
from oracle import ref_vit as RV
from oracle import ref_whisper as RW
from synthweights import bf16_round_, fill_module, synth_input, synth_tokens

torch.set_grad_enabled(False)
TOL = dict(rtol=2e-5, atol=2e-5)


def sd_of(m, seed):
    fill_module(m, seed)
    return {k: v.clone() for k, v in m.state_dict().items()}


def digest(t):
    f = t.double().flatten()
    w = 1.0 + (torch.arange(f.numel(), dtype=torch.float64) % 251) / 251.0
    return torch.tensor([f.sum().item(), f.abs().sum().item(), (f * w).sum().item()], dtype=torch.float64)


def assert_digest(t, want, rel=1e-5):
    got = digest(t)
    scale = want[1].abs()  # sum |x|
    assert ((got - want).abs() <= rel * scale).all(), (got, want)


# ------------------------------------------------------------------ transformer blocks
def test_blocks(golden):
    from pytorch_models.transformer import Decoder, DecoderLayer, Encoder, EncoderLayer

    g = golden("blocks")
    d = 64
    x = synth_input("blk_x", (2, 10, d), 1)
    mem = synth_input("blk_mem", (2, 7, d), 1)
    for pre in (True, False):
        for eps in (1e-5, 1e-6):
            sd = sd_of(EncoderLayer(d, pre_norm=pre, norm_eps=eps), 11)
            got = RT.encoder_layer(sd, "", 1, x, pre_norm=pre, eps=eps)
            torch.testing.assert_close(got, g[f"enc_pre{int(pre)}_eps{eps}"], **TOL)
            sd = sd_of(DecoderLayer(d, cross_attn=True, pre_norm=pre, norm_eps=eps), 12)
            got = RT.decoder_layer(sd, "", 1, x, mem, pre_norm=pre, eps=eps)
            torch.testing.assert_close(got, g[f"dec_pre{int(pre)}_eps{eps}"], **TOL)
    sd = sd_of(DecoderLayer(d, cross_attn=False), 13)
    torch.testing.assert_close(RT.decoder_layer(sd, "", 1, x), g["dec_nocross"], **TOL)
    for act in ("gelu", "approximate_gelu", "relu", "silu"):
        sd = sd_of(EncoderLayer(d, act=act), 14)
        torch.testing.assert_close(RT.encoder_layer(sd, "", 1, x, act=act), g[f"enc_act_{act}"], **TOL)
    sd = sd_of(Encoder(3, 128, n_heads=2), 15)
    torch.testing.assert_close(RT.encoder(sd, "", 2, synth_input("blk_x128", (2, 9, 128), 1)), g["encoder3"], **TOL)
    sd = sd_of(Decoder(2, 128, cross_attn=True), 16)
    got = RT.decoder(sd, "", 2, synth_input("blk_x128", (2, 9, 128), 1), synth_input("blk_mem128", (2, 5, 128), 1))
    torch.testing.assert_close(got, g["decoder2"], **TOL)


def test_mha_variants(golden):
    from pytorch_models.transformer import MHA

    g = golden("mha")
    d = 64
    q = synth_input("mha_q", (2, 6, d), 2)
    k = synth_input("mha_k", (2, 9, d), 2)
    v = synth_input("mha_v", (2, 9, d), 2)
    bias = synth_input("mha_bias", (2, 1, 6, 9), 2)
    sd = sd_of(MHA(d), 21)
    torch.testing.assert_close(RT.mha(sd, "", 1, q), g["default_q"], **TOL)
    sd = sd_of(MHA(d, n_heads=4), 22)
    torch.testing.assert_close(RT.mha(sd, "", 4, q), g["h4_q"], **TOL)
    torch.testing.assert_close(RT.mha(sd, "", 4, q, k), g["h4_qk"], **TOL)
    torch.testing.assert_close(RT.mha(sd, "", 4, q, k, v), g["h4_qkv"], **TOL)
    torch.testing.assert_close(RT.mha(sd, "", 4, q, k, v, attn_bias=bias), g["h4_bias"], **TOL)
    keep = torch.ones(2, 1, 1, 9, dtype=torch.bool)  # key-padding mask, broadcast over heads and queries
    keep[0, ..., 6:] = False
    keep[1, ..., 8:] = False
    torch.testing.assert_close(RT.mha(sd, "", 4, q, k, v, attn_bias=keep), g["h4_keypad_bool"], **TOL)
    add = torch.zeros(2, 1, 1, 9).masked_fill(~keep, float("-inf"))
    torch.testing.assert_close(RT.mha(sd, "", 4, q, k, v, attn_bias=add), g["h4_keypad_add"], **TOL)
    torch.testing.assert_close(g["h4_keypad_add"], g["h4_keypad_bool"], **TOL)
    torch.testing.assert_close(RT.mha(sd, "", 4, q, causal=True), g["h4_causal"], **TOL)
    torch.testing.assert_close(RT.mha(sd, "", 4, q, k, causal=True), g["h4_causal_rect"], **TOL)
    torch.testing.assert_close(RT.mha(sd, "", 4, q[0]), g["h4_unbatched"], **TOL)
    m = MHA(d, n_heads=2, head_dim=16)
    assert m.q_proj.weight.shape == (32, d) and m.out_proj.weight.shape == (d, 32)
    torch.testing.assert_close(RT.mha(sd_of(m, 23), "", 2, q), g["h2hd16_q"], **TOL)
    m = MHA(d, head_dim=32, bias=False)
    assert m.n_heads == 2 and m.q_proj.bias is None
    torch.testing.assert_close(RT.mha(sd_of(m, 24), "", 2, q, k), g["hd32_nobias"], **TOL)


def test_sdpa_causal_is_top_left_aligned(golden):
    """L_q = 1 against S_k = 5 with causal=True sees key 0 only."""
    g = golden("sdpa_alignment")
    q = synth_input("f3_q", (1, 1, 1, 8), 3)
    k = synth_input("f3_k", (1, 1, 5, 8), 3)
    v = synth_input("f3_v", (1, 1, 5, 8), 3)
    out = RT.sdpa(q, k, v, causal=True)
    torch.testing.assert_close(out, g["out"], **TOL)
    torch.testing.assert_close(out[:, :, 0], g["v0"], **TOL)


# ------------------------------------------------------------------ ViT
def test_vit(golden):
    from pytorch_models.image import ViT

    g = golden("vit")
    m = ViT.from_google("Ti/16")
    sd = sd_of(m, 31)
    geo = RV.geometry_from_google("Ti/16")
    x = synth_input("vit_ti", (1, 3, 224, 224), 31)
    torch.testing.assert_close(RV.forward(sd, geo, x), g["ti16_b1"], **TOL)
    t = RV.tokens(sd, x)
    torch.testing.assert_close(t[0, :5, :16], g["ti16_tokens_slice"], **TOL)
    assert_digest(t, g["ti16_tokens_digest"])
    # resize_pe(256) + forward at 256 (tests/image/test_vit.py:21-26 of the reference)
    pe256 = RV.resize_pe(sd["pe"], 16, 256)
    torch.testing.assert_close(pe256, g["ti16_pe256"], **TOL)
    sd256 = dict(sd, pe=pe256)
    torch.testing.assert_close(RV.forward(sd256, geo, synth_input("vit_ti256", (1, 3, 256, 256), 31)), g["ti16_b1_256"], **TOL)
    m.resize_pe(256)  # the product's own resize_pe (host-side utility) agrees too
    torch.testing.assert_close(m.pe.detach(), g["ti16_pe256"], **TOL)


def test_vit_b16_per_sample_loop(golden):
    """F1: batch > 1 with a cls token == stack of the reference's batch-1 results."""
    from pytorch_models.image import ViT

    g = golden("vit")
    sd = sd_of(ViT.from_google("B/16"), 32)
    xb = synth_input("vit_b", (4, 3, 224, 224), 32)
    got = RV.forward(sd, RV.geometry_from_google("B/16"), xb)  # one batched call
    torch.testing.assert_close(got, g["b16_first4"], **TOL)


def test_vit_siglip_and_poolers(golden):
    from pytorch_models.image import ViT

    g = golden("vit")
    sd = sd_of(ViT.from_google("B/16_siglip"), 33)
    got = RV.forward(sd, RV.geometry_from_google("B/16_siglip"), synth_input("vit_bs", (2, 3, 224, 224), 33))
    torch.testing.assert_close(got, g["b16_siglip_b2"], **TOL)
    sd = sd_of(ViT(2, 128, 2, 16, img_size=64, pool_type="gap"), 36)
    geo = RV.ViTGeometry(2, 128, 2, 16, 64, True, "gap")
    torch.testing.assert_close(RV.forward(sd, geo, synth_input("vit_gap", (1, 3, 64, 64), 36)), g["tiny_gap_b1"], **TOL)
    sd = sd_of(ViT(2, 128, 2, 16, img_size=64, cls_token=False, pool_type="gap"), 37)
    geo = RV.ViTGeometry(2, 128, 2, 16, 64, False, "gap")
    torch.testing.assert_close(RV.forward(sd, geo, synth_input("vit_gap3", (3, 3, 64, 64), 37)), g["tiny_gap_nocls_b3"], **TOL)


def test_vit_large_configs(golden):
    from pytorch_models.image import ViT

    g = golden("vit")
    sd = sd_of(ViT.from_google("L/16_siglip", img_size=384), 34)
    geo = RV.geometry_from_google("L/16_siglip", img_size=384)
    got = RV.forward(sd, geo, synth_input("vit_ls", (2, 3, 384, 384), 34))
    torch.testing.assert_close(got, g["l16_siglip384_b2"], **TOL)
    sd = sd_of(ViT.from_facebook("S/14_dinov2"), 35)
    geo = RV.geometry_from_facebook("S/14_dinov2")
    assert geo.img_size == 518
    got = RV.forward(sd, geo, synth_input("vit_dv2", (1, 3, 518, 518), 35))
    torch.testing.assert_close(got, g["s14_dinov2_b1"], **TOL)


# ------------------------------------------------------------------ audio front end
def test_mel_filters(golden):
    g = golden("audio")
    for n in (80, 128):
        f = RS.mel_filters(n, 400, 16000)
        # the reference builds the bank in fp32 (spectrogram.py:23-34), the oracle in fp64: band edges move by ~1e-7
        torch.testing.assert_close(f, g[f"filters{n}"], rtol=1e-4, atol=5e-7)
        assert abs(int((f > 1e-6).sum()) - int((g[f"filters{n}"] > 1e-6).sum())) <= 2


@pytest.mark.parametrize("dft", ["matmul", "rfft"])
def test_spectrogram_and_logmel(golden, dft):
    g = golden("audio")
    x1 = synth_input("wave_1s", (16000,), 41)
    spec = RS.power_spectrogram(x1, 400, 160, dft)
    assert spec.shape == (201, 101)
    torch.testing.assert_close(spec, g["spec_1s"], rtol=2e-5, atol=2e-4)  # values reach ~1e3: atol scaled
    torch.testing.assert_close(RS.mel_spectrogram(x1, 400, 160, 80, 16000, dft), g["mel_1s"], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(RS.whisper_log_mel(x1, 80, dft), g["logmel_1s"], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(RS.whisper_log_mel(x1, 128, dft), g["logmel128_1s"], rtol=1e-5, atol=2e-5)


def test_logmel_30s_and_batch(golden):
    g = golden("audio")
    x30 = synth_input("wave_30s", (2, 480000), 42, scale=0.1)
    lm = RS.whisper_log_mel(x30, 80, "rfft")
    assert lm.shape == (2, 80, 3000)
    torch.testing.assert_close(lm[:, ::8, ::100], g["logmel_30s_slice"], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(lm[0, :, :8], g["logmel_30s_head"], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(lm[1, :, -8:], g["logmel_30s_tail"], rtol=1e-5, atol=2e-5)
    assert_digest(lm, g["logmel_30s_digest"], rel=2e-5)
    # reference tests/audio2text/test_whisper.py:57-65: per-sample max (F4)
    xb = synth_input("wave_batch", (4, 16000), 43)
    xb[0] += 0.5
    got = RS.whisper_log_mel(xb)
    torch.testing.assert_close(got, g["logmel_batch"], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(got, torch.stack([RS.whisper_log_mel(xb[i]) for i in range(4)]), rtol=0, atol=1e-6)
    xs = synth_input("wave_half_silent", (16000,), 44)
    xs[8000:] = 0
    torch.testing.assert_close(RS.whisper_log_mel(xs), g["logmel_half_silent"], rtol=1e-5, atol=2e-5)


# ------------------------------------------------------------------ Whisper
def test_whisper_smoke_shapes(golden):
    from pytorch_models.audio2text import Whisper, WhisperDecoder, WhisperEncoder

    g = golden("whisper")
    vocab, L, d = 100, 2, 64
    mel = synth_input("w_mel16", (2, 80, 16), 51)
    toks = synth_tokens("w_tok32", (2, 32), vocab, 52)
    sd = sd_of(WhisperEncoder(L, d), 51)
    torch.testing.assert_close(RW.encoder(sd, "", mel), g["smoke_encoder"], **TOL)
    sd = sd_of(WhisperDecoder(vocab, L, d), 52)
    torch.testing.assert_close(RW.decoder(sd, "", toks, synth_input("w_mem16", (2, 16, d), 52)), g["smoke_decoder"], **TOL)
    sd = sd_of(Whisper(vocab, L, d), 53)
    torch.testing.assert_close(RW.forward(sd, mel, toks), g["smoke_whisper"], **TOL)


def test_whisper_tiny_logits(golden):
    from pytorch_models.audio2text import Whisper

    g = golden("whisper")
    sd = sd_of(Whisper.from_openai("tiny"), 54)
    mel = synth_input("w_mel3000", (1, 80, 3000), 54)
    toks = synth_tokens("w_tok200", (1, 32), 200, 54)
    memory = RW.encoder(sd, "encoder.", mel)
    torch.testing.assert_close(memory[0, ::100, ::32], g["tiny_memory_slice"], **TOL)
    assert_digest(memory, g["tiny_memory_digest"])
    logits = RW.decoder(sd, "decoder.", toks, memory)
    torch.testing.assert_close(logits[0, :, :128], g["tiny_logits_slice"], rtol=5e-5, atol=5e-5)
    torch.testing.assert_close(logits.max(-1).values, g["tiny_logits_max"], rtol=5e-5, atol=5e-5)
    assert torch.equal(logits.argmax(-1), g["tiny_logits_argmax"])


@pytest.mark.parametrize("tag,seed", [("tiny", 55), ("base", 56)])
def test_whisper_greedy_ids(golden, tag, seed):
    """Greedy ids of the reference's full-prefix recompute, bit-exact, through log-mel + encoder + decode;
    and the KV-cached restatement (the algorithm the HIP path implements) gives the same ids."""
    from pytorch_models.audio2text import Whisper

    g = golden("whisper")
    sd = sd_of(Whisper.from_openai(tag), seed)
    wave = synth_input(f"w_wave_{tag}", (2, 480000), seed, scale=0.1)
    memory = RW.encoder(sd, "encoder.", RS.whisper_log_mel(wave, 80, "rfft"))
    torch.testing.assert_close(memory[:, ::100, ::32], g[f"greedy_{tag}_memory_slice"], **TOL)
    prompt = synth_tokens(f"w_prompt_{tag}", (2, 4), 51865, seed)
    want, margins = g[f"greedy_{tag}_tokens"], g[f"greedy_{tag}_margins"]
    toks_c, marg_c = RW.greedy_cached(sd, "decoder.", prompt, memory, 32)
    assert torch.equal(toks_c, want), (toks_c != want).nonzero()
    torch.testing.assert_close(marg_c, margins, rtol=0, atol=2e-4)
    toks_r, _ = RW.greedy_recompute(sd, "decoder.", prompt, memory, 8)
    assert torch.equal(toks_r, want[:, :12])


@pytest.mark.parametrize("tag,seed", [("tiny", 55), ("base", 56)])
@pytest.mark.parametrize("kind", ["f", "r"])
def test_greedy_ids_224_tokens_bit_exact(golden, tag, seed, kind):
    """The full-length decode of BASELINE configs[2] (prompt 4, 224 new tokens; make_golden.py ran the reference's
    full-prefix loop): the oracle's KV-cached loop gives the same ids bit for bit, on the reference's plain fp32 weights
    ("f") and on bf16-representable weights ("r": the weights a bf16 model on the GPU holds)."""
    from pytorch_models.audio2text import Whisper

    g = golden("whisper")
    m = Whisper.from_openai(tag)
    fill_module(m, seed)
    if kind == "r":
        bf16_round_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    wave = synth_input(f"w_wave_{tag}", (2, 480000), seed, scale=0.1)
    memory = RW.encoder(sd, "encoder.", RS.whisper_log_mel(wave, 80, "rfft"))
    prompt = synth_tokens(f"w_prompt_{tag}", (2, 4), 51865, seed)
    toks, margins = RW.greedy_cached(sd, "decoder.", prompt, memory, 224)
    assert torch.equal(toks, g[f"greedy224{kind}_{tag}_tokens"])
    torch.testing.assert_close(margins, g[f"greedy224{kind}_{tag}_margins"], rtol=1e-3, atol=2e-5)
