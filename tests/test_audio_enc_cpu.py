"""CPU: the wav2vec2 family of SURVEY.md 8(f) row 2 - the oracle (oracle/ref_audio.py) against vectors captured from the
reference (tests/golden/audio_enc.npz, made by tests/golden/make_golden.py audio_enc), constructor contracts, and the
HF weight loaders against digests of what the reference's loaders produce (tests/golden/audio_converters.json)."""
import io
import json
import os
from contextlib import redirect_stdout

import pytest
import torch

import ckpt_synth as C
from oracle import ref_audio as RA
from synthweights import fill_module, synth_input

torch.set_grad_enabled(False)
TOL = dict(rtol=2e-5, atol=2e-5)
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "audio_converters.json")))


def cases():
    from pytorch_models.audio import SEW, Data2VecAudio, Wav2Vec2

    return dict(
        w2v_d64=(lambda: Wav2Vec2(2, 64), 82, lambda sd, x: RA.wav2vec2(sd, x), RA.STEM_STRIDES, False),
        w2v_legacy_post_d128=(lambda: Wav2Vec2(2, 128, stem_bias=False, stem_legacy=True, pre_norm=False), 83,
                              lambda sd, x: RA.wav2vec2(sd, x, pre_norm=False, legacy=True), RA.STEM_STRIDES, True),
        d2v_d128=(lambda: Data2VecAudio(2, 128), 84, RA.data2vec_audio, RA.STEM_STRIDES, False),
        sew_d128=(lambda: SEW(2, 128), 85, RA.sew, RA.SEW_STRIDES, True),
    )


def sd_of(m, seed):
    fill_module(m, seed)
    return {k: v.clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("name", ["w2v_d64", "w2v_legacy_post_d128", "d2v_d128", "sew_d128"])
def test_oracle_matches_the_reference(golden, name):
    g = golden("audio_enc")
    make, seed, fwd, strides, legacy = cases()[name]
    sd = sd_of(make(), seed)
    x = synth_input("w2v_x", (2, 6400), 81)
    feat = RA.feature_encoder(sd, "feature_encoder.", x, strides, legacy)
    torch.testing.assert_close(feat[..., ::8], g[name + "_feat_s8"], **TOL)
    torch.testing.assert_close(RA._project(sd, feat), g[name + "_proj"], rtol=2e-5, atol=1e-4)
    torch.testing.assert_close(fwd(sd, x), g[name], rtol=2e-5, atol=1e-4)
    if name == "sew_d128":
        assert g[name].shape[1] == 19 and g[name][:, -1].abs().max() == 0  # odd frame count: trailing zero frame
        torch.testing.assert_close(fwd(sd, x[:, :6080]), g["sew_d128_even"], rtol=2e-5, atol=1e-4)


def test_constructors_mirror_the_reference():
    from pytorch_models.audio import SEW, Data2VecAudio, Wav2Vec2

    m = Wav2Vec2(2, 64)
    assert [b[0].kernel_size[0] for b in m.feature_encoder] == [10, 3, 3, 3, 3, 2, 2]
    assert [b[0].stride[0] for b in m.feature_encoder] == [5, 2, 2, 2, 2, 2, 2]
    assert len(m.proj) == 2 and m.pe_conv[1].groups == 16 and m.pe_conv[1].kernel_size == (128,) and m.pe_conv[0].padding == (64, 63)
    assert len(Wav2Vec2(1, 512).proj) == 1  # no projection when the stem already has d_model channels (wav2vec2.py:67-68)
    legacy = Wav2Vec2(1, 64, stem_bias=False, stem_legacy=True)
    assert isinstance(legacy.feature_encoder[0][2], torch.nn.InstanceNorm1d) and isinstance(legacy.feature_encoder[1][2], torch.nn.Identity)
    assert legacy.feature_encoder[0][0].bias is None
    d2v = Data2VecAudio(1, 64)
    assert len(d2v.pe_conv) == 5 and d2v.pe_conv[0][0].kernel_size == (19,) and d2v.pe_conv[0][1].weight is None and not d2v.pre_norm
    sew = SEW(1, 64)
    assert len(sew.feature_encoder) == 13 and sew.pe_conv[1].stride == (2,) and sew.upsample[0].out_features == 128
    with pytest.raises(AssertionError):
        SEW(1, 64, stem_legacy=False)
    with pytest.raises(NotImplementedError, match="no network"):
        Wav2Vec2.from_hf("facebook/wav2vec2-base")
    cfg = dict(hidden_size=128, num_attention_heads=2, num_hidden_layers=3, conv_bias=False, feat_extract_norm="group",
               do_stable_layer_norm=False)
    m = Wav2Vec2.from_hf("facebook/wav2vec2-base", config=cfg)
    assert len(m.layers) == 3 and not m.pre_norm and isinstance(m.feature_encoder[0][2], torch.nn.InstanceNorm1d)
    with pytest.raises(RuntimeError, match="HIP devices only"):
        m(torch.zeros(1, 400))


def test_names_match_the_reference_state_dict(golden):
    """The digests were taken from the reference's own state_dict(): same keys means its checkpoints load here."""
    from pytorch_models.audio import SEW, Data2VecAudio, Wav2Vec2

    assert sorted(Wav2Vec2(2, 128).state_dict()) == sorted(GOLD["hf_wav2vec2"])
    assert sorted(Wav2Vec2(2, 128, stem_bias=False, stem_legacy=True, pre_norm=False).state_dict()) == sorted(GOLD["hf_wav2vec2_base"])
    assert sorted(Data2VecAudio(2, 128).state_dict()) == sorted(GOLD["hf_data2vec"])
    assert sorted(SEW(2, 128).state_dict()) == sorted(GOLD["hf_sew"])


@pytest.mark.parametrize("name", ["hf_wav2vec2", "hf_wav2vec2_base", "hf_data2vec", "hf_sew"])
def test_hf_loaders_match_the_reference(name):
    from pytorch_models.audio import SEW, Data2VecAudio, Wav2Vec2

    make, kw = dict(
        hf_wav2vec2=(lambda: Wav2Vec2(2, 128), dict(kind="wav2vec2", legacy=False, stem_bias=True, pe_kernel=128)),
        hf_wav2vec2_base=(lambda: Wav2Vec2(2, 128, stem_bias=False, stem_legacy=True, pre_norm=False),
                          dict(kind="wav2vec2", legacy=True, stem_bias=False, pe_kernel=128)),
        hf_data2vec=(lambda: Data2VecAudio(2, 128), dict(kind="data2vec", legacy=False, stem_bias=False, pe_kernel=19)),
        hf_sew=(lambda: SEW(2, 128), dict(kind="sew", legacy=True, stem_bias=True, pe_kernel=31)),
    )[name]
    m = make()
    kind = kw.pop("kind")
    with redirect_stdout(io.StringIO()) as so:
        m.load_hf_state_dict(C.hf_wav2vec2(kind, 2, 128, m.STEM_DIMS, m.STEM_KERNELS, seed=86, **kw))
    assert "dict_keys([])" in so.getvalue()  # every upstream key consumed
    got = C.state_digest(m.state_dict())
    assert sorted(got) == sorted(GOLD[name])
    for k, want in GOLD[name].items():
        w, gt = torch.tensor(want, dtype=torch.float64), torch.tensor(got[k], dtype=torch.float64)
        assert ((w - gt).abs() <= 1e-6 * w[1].abs() + 1e-9).all(), (k, want, got[k])
