"""The CPU-device forms (pytorch_models/_cpu.py: plain torch, selected only when the input AND the parameters are on the CPU)
against the reference's OWN fp32 vectors at the reference's OWN tolerances - BASELINE.json configs[0] ("ViT-Ti/16 augreg
forward, batch=1, 224x224 on the repo's CPU path") and the rest of the hot path's classes: blocks.npz / mha.npz / vit.npz at
rtol = atol = 2e-5 (reference tests/image/test_vit.py:45), whisper.npz at 5e-5 (tests/audio2text/test_whisper.py:45), audio.npz
at the spectrogram tests' 2e-5 / assert_close defaults.  No GPU, no oracle/, no HIP library call: the last test checks that a
CPU forward never touches libpm_mi355x.so and that mixed placements still raise."""
import pytest
import torch

from synthweights import fill_module, synth_input, synth_tokens

torch.set_grad_enabled(False)
REF = dict(rtol=2e-5, atol=2e-5)


def test_config0_vit_ti16_batch1_on_the_cpu(golden):
    from pytorch_models.image import ViT

    g = golden("vit")
    m = ViT.from_google("Ti/16").eval()
    fill_module(m, 31)
    got = m(synth_input("vit_ti", (1, 3, 224, 224), 31))
    assert got.device.type == "cpu" and got.dtype == torch.float32 and got.shape == (1, 192)
    torch.testing.assert_close(got, g["ti16_b1"], **REF)
    m.resize_pe(256)
    torch.testing.assert_close(m(synth_input("vit_ti256", (1, 3, 256, 256), 31)), g["ti16_b1_256"], **REF)


def test_vit_variants_on_the_cpu(golden):
    from pytorch_models.image import ViT

    g = golden("vit")
    m = ViT.from_google("B/16_siglip").eval()  # MAP pooling, no cls token
    fill_module(m, 33)
    torch.testing.assert_close(m(synth_input("vit_bs", (2, 3, 224, 224), 33)), g["b16_siglip_b2"], **REF)
    m = ViT.from_google("B/16").eval()  # batch > 1 with a cls token: the stack of the reference's batch-1 results
    fill_module(m, 32)
    torch.testing.assert_close(m(synth_input("vit_b", (4, 3, 224, 224), 32)[:2]), g["b16_first4"][:2], **REF)


def test_blocks_on_the_cpu(golden):
    from pytorch_models.transformer import Decoder, DecoderLayer, Encoder, EncoderLayer

    g = golden("blocks")
    x, mem = synth_input("blk_x", (2, 10, 64), 1), synth_input("blk_mem", (2, 7, 64), 1)
    for pre in (True, False):
        for eps in (1e-5, 1e-6):
            m = EncoderLayer(64, pre_norm=pre, norm_eps=eps).eval()
            fill_module(m, 11)
            torch.testing.assert_close(m(x), g[f"enc_pre{int(pre)}_eps{eps}"], **REF)
            m = DecoderLayer(64, cross_attn=True, pre_norm=pre, norm_eps=eps).eval()
            fill_module(m, 12)
            torch.testing.assert_close(m(x, mem), g[f"dec_pre{int(pre)}_eps{eps}"], **REF)
    for act in ("gelu", "approximate_gelu", "relu", "silu"):
        m = EncoderLayer(64, act=act).eval()
        fill_module(m, 14)
        torch.testing.assert_close(m(x), g[f"enc_act_{act}"], **REF)
    m = Encoder(3, 128, n_heads=2).eval()
    fill_module(m, 15)
    torch.testing.assert_close(m(synth_input("blk_x128", (2, 9, 128), 1)), g["encoder3"], **REF)
    m = Decoder(2, 128, cross_attn=True).eval()
    fill_module(m, 16)
    torch.testing.assert_close(m(synth_input("blk_x128", (2, 9, 128), 1), synth_input("blk_mem128", (2, 5, 128), 1)), g["decoder2"], **REF)


def test_whisper_tiny_logits_on_the_cpu(golden):
    """tests/audio2text/test_whisper.py:39-45's shapes on "tiny": memory and logits at 5e-5, arg-max ids equal."""
    from pytorch_models.audio2text import Whisper

    g = golden("whisper")
    w = Whisper.from_openai("tiny").eval()
    fill_module(w, 54)
    memory = w.encoder(synth_input("w_mel3000", (1, 80, 3000), 54))
    torch.testing.assert_close(memory[0, ::100, ::32], g["tiny_memory_slice"], rtol=5e-5, atol=5e-5)
    logits = w.decoder(synth_tokens("w_tok200", (1, 32), 200, 54), memory)
    torch.testing.assert_close(logits[0, :, :128], g["tiny_logits_slice"], rtol=5e-5, atol=5e-5)
    assert torch.equal(logits.argmax(-1), g["tiny_logits_argmax"])


def test_front_end_on_the_cpu(golden):
    from pytorch_models.audio.spectrogram import MelSpectrogram, Spectrogram
    from pytorch_models.audio2text import WhisperPreprocessor

    g = golden("audio")
    wave = synth_input("wave_1s", (16000,), 41)  # tests/golden/make_golden.py g_audio
    torch.testing.assert_close(Spectrogram(400, 160)(wave), g["spec_1s"], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(MelSpectrogram(400, 160, 80, 16000)(wave), g["mel_1s"], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(WhisperPreprocessor()(wave), g["logmel_1s"])


def test_cpu_forward_never_touches_the_hip_library_and_mixed_placements_raise(monkeypatch):
    from pytorch_models import _hip
    from pytorch_models.transformer import EncoderLayer

    def boom():
        raise AssertionError("a CPU forward called into libpm_mi355x.so")

    monkeypatch.setattr(_hip, "lib", boom)
    monkeypatch.setattr(_hip.ops, "lib", boom)
    m = EncoderLayer(64).eval()
    fill_module(m, 11)
    m(synth_input("blk_x", (2, 10, 64), 1))  # plain torch end to end
    if torch.cuda.is_available():  # input on the GPU, parameters on the CPU: refused, not silently computed somewhere
        with pytest.raises(RuntimeError):
            m(synth_input("blk_x", (2, 10, 64), 1).cuda())
