"""CPU: the product constructors reproduce the reference's state_dict layout (names and shapes captured
in tests/golden/geometry.json from the reference import) and its tag / error behaviour
(/root/reference pytorch_models/image/vit.py:96-119,202-239, audio2text/whisper.py:65-94)."""
import json
import os

import pytest
import torch

from pytorch_models.audio.spectrogram import MelSpectrogram, Spectrogram, get_mel_filters
from pytorch_models.audio2text import Whisper, WhisperDecoder, WhisperEncoder, WhisperPreprocessor
from pytorch_models.image import ViT
from pytorch_models.transformer import MHA, MLP, Decoder, DecoderLayer, Encoder, EncoderLayer

GEO = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "geometry.json")))


def shapes(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


@pytest.mark.parametrize("tag", ["Ti/16", "B/16", "B/16_siglip"])
def test_vit_google(tag):
    assert shapes(ViT.from_google(tag)) == GEO["google:" + tag]


def test_vit_google_siglip384():
    with torch.device("meta"):
        m = ViT.from_google("L/16_siglip", img_size=384)
    assert shapes(m) == GEO["google:L/16_siglip@384"]
    assert m.cls_token is None and type(m.pooler).__name__ == "MHAPooling"


@pytest.mark.parametrize("tag", ["S/16_deit3", "S/16_dino", "S/14_dinov2"])
def test_vit_facebook(tag):
    assert shapes(ViT.from_facebook(tag)) == GEO["facebook:" + tag]


@pytest.mark.parametrize("tag", ["tiny", "tiny.en", "base", "large-v3"])
def test_whisper_openai(tag):
    with torch.device("meta"):
        m = Whisper.from_openai(tag)
    assert shapes(m) == GEO["openai:" + tag]


def test_whisper_base_is_8_layers_like_the_reference():
    with torch.device("meta"):
        m = Whisper.from_openai("base")
    assert len(m.encoder.layers) == 8 and len(m.decoder.layers) == 8


def test_preprocessor_buffers():
    p = WhisperPreprocessor()
    assert shapes(p) == GEO["preprocessor"]  # filters persistent, window not (spectrogram.py:12,41)
    assert p.window.shape == (400,) and WhisperPreprocessor("large-v3").filters.shape == (128, 201)
    assert isinstance(p, MelSpectrogram) and isinstance(p, Spectrogram)


def test_error_behaviour():
    with pytest.raises(KeyError):
        ViT.from_google("XXL/16")
    with pytest.raises(KeyError):
        Whisper.from_openai("huge")
    with pytest.raises(ValueError):
        ViT.from_facebook("S/16_foo")
    with pytest.raises(AssertionError):
        ViT(1, 64, 1, 16, img_size=100)
    with pytest.raises(KeyError):
        MLP(8, 16, act="nope")
    with pytest.raises(KeyError):
        ViT(1, 64, 1, 16, pool_type="nope")


def test_mha_head_resolution():
    m = MHA(192)
    assert (m.n_heads, m.head_dim) == (3, 64)
    m = MHA(192, n_heads=6)
    assert (m.n_heads, m.head_dim) == (6, 32)
    m = MHA(512, head_dim=32)
    assert (m.n_heads, m.head_dim) == (16, 32)
    m = MHA(512, n_heads=6, head_dim=64)  # n_heads * head_dim < d_model (T5-small style)
    assert m.q_proj.weight.shape == (384, 512) and m.out_proj.weight.shape == (512, 384)
    assert m.dropout == 0.0


def test_layer_structure():
    e = EncoderLayer(64)
    assert e.ca is None and e.ca_norm is None and e.pre_norm
    d = DecoderLayer(64, cross_attn=True, pre_norm=False, norm_eps=1e-12)
    assert d.ca is not None and d.ca_norm.eps == 1e-12 and not d.pre_norm
    assert len(Encoder(3, 64)) == 3 and len(Decoder(2, 64, cross_attn=True)) == 2
    assert [n for n, _ in MLP(8, 32).named_children()] == ["linear1", "act", "linear2", "dropout"]
    assert isinstance(e.sa, MHA) and isinstance(e.mlp.linear1, torch.nn.Linear) and isinstance(e.sa_norm, torch.nn.LayerNorm)


def test_state_dict_roundtrip_and_derived_cache_invalidation():
    m = MHA(64, n_heads=1)
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    w1, _ = m._pack("qkv")  # the packed projection is always bf16 (an fp32 model runs through bf16 copies of its weights)
    assert w1.dtype == torch.bfloat16 and torch.equal(w1[:64], sd["q_proj.weight"].to(torch.bfloat16)) and w1.shape == (192, 64)
    assert m._pack("qkv")[0] is w1  # cached
    with torch.no_grad():
        m.k_proj.weight.copy_(torch.zeros(64, 64))  # in-place load (what the converters do) must invalidate
    w2, _ = m._pack("qkv")
    assert w2 is not w1 and w2[64:128].abs().sum() == 0
    assert "_pm_derived" not in m.state_dict()


def test_cpu_tensors_take_the_cpu_forms_and_never_the_hip_library(monkeypatch):
    """A module and its input both on the CPU compute in plain torch (pytorch_models/_cpu.py: BASELINE configs[0]); the HIP
    library is not touched, and - the other half of "no silent fallback" - the families whose kernels are HIP-only still
    refuse a CPU model.  (HIP tensors with a missing library raise in pytorch_models._hip.lib: tests/test_abi.py.)"""
    from pytorch_models import _hip

    def boom():
        raise AssertionError("a CPU forward called into libpm_mi355x.so")

    monkeypatch.setattr(_hip, "lib", boom)
    monkeypatch.setattr(_hip.ops, "lib", boom)
    y = EncoderLayer(64)(torch.zeros(1, 4, 64))
    assert y.shape == (1, 4, 64) and y.device.type == "cpu"
    assert ViT(1, 64, 1, 16, img_size=32)(torch.zeros(1, 3, 32, 32)).shape == (1, 64)
    from pytorch_models.transformer import require_bf16_params

    with pytest.raises(RuntimeError, match="HIP devices only"):
        require_bf16_params(EncoderLayer(64), "T5")


def test_mel_filters_match_reference_golden(golden):
    g = golden("audio")
    torch.testing.assert_close(get_mel_filters(80, 400, 16000), g["filters80"], rtol=1e-4, atol=5e-7)
    torch.testing.assert_close(get_mel_filters(128, 400, 16000), g["filters128"], rtol=1e-4, atol=5e-7)


def test_whisper_shallow_decoder_keeps_parameter_names():
    from pytorch_models.audio2text import Whisper

    full, distil = Whisper(100, 3, 64), Whisper(100, 3, 64, n_decoder_layers=1)
    assert len(distil.encoder.layers) == 3 and len(distil.decoder.layers) == 1
    assert set(distil.state_dict()) < set(full.state_dict())  # a strict subset: the decoder layers that are gone


def test_graphed_forward_refuses_cpu_inputs():
    import torch

    from pytorch_models.graph import GraphedForward

    with pytest.raises(RuntimeError, match="HIP device"):
        GraphedForward(torch.nn.Identity(), torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="HIP device"):
        GraphedForward(torch.nn.Identity())
