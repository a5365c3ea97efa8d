"""GPU parity of the wav2vec2 family through the C ABI: the stem / regrouping / pooling kernels
against the fp32 oracle (oracle/ref_audio.py), conv layers as strided-window GEMMs, and Wav2Vec2 / Data2VecAudio / SEW
end to end against the oracle on the same bf16-rounded weights and against the reference's own vectors
(tests/golden/audio_enc.npz).

Tolerances: kernels that compute in fp32 and round once to bf16: |err| <= 4e-3 |want| + 1e-3 (one bf16 ulp is 3.9e-3
relative); GEMM-backed pieces and whole models: rel-L2 <= 2e-2 vs the oracle, 3e-2 vs the reference golden (fp32
weights), as in test_hip_blocks.py."""
import pytest
import torch

from oracle import ref_audio as RA
from oracle import ref_transformer as RT
from synthweights import bf16_round_, fill_module, synth_input, synth_tensor

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rel(got, want):
    got, want = got.float().cpu(), want.float()
    return ((got - want).norm() / want.norm()).item()


def once_rounded(got, want):
    torch.testing.assert_close(got.float().cpu(), want, rtol=4e-3, atol=1e-3)


@pytest.mark.parametrize("norm", ["none", "layer", "instance"])
@pytest.mark.parametrize("C0,L,bias", [(512, 4000, True), (64, 1603, False), (8, 10, True)])
def test_stem0_conv_norm_gelu(norm, C0, L, bias):
    from pytorch_models._hip import ops

    x = synth_input("stem0_x", (3, L), 5)
    w = synth_tensor("stem0.weight", (C0, 1, 10), 5)
    b = synth_tensor("stem0.bias", (C0,), 5) if bias else None
    g, be = synth_tensor("stem0.norm_weight", (C0,), 5) + 1.0, synth_tensor("stem0.norm.bias", (C0,), 5)
    h = RA.conv1d_tl(w, b, x[:, :, None], 5)
    if norm != "none":
        h = RA._norm_free(h, 1 if norm == "instance" else 2, 1e-5) * g + be
    want = RT.activation(h, "gelu")
    dev = lambda t: None if t is None else t.cuda()
    got = ops.w2v_stem0(x.cuda(), w.view(C0, 10).cuda(), dev(b), norm, dev(g) if norm != "none" else None,
                        dev(be) if norm != "none" else None, 1e-5, 5)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    if L == 10 and norm == "instance":  # a single step: variance 0, everything collapses onto beta
        want = RT.activation(be.expand_as(want), "gelu")
    once_rounded(got, want)
    again = ops.w2v_stem0(x.cuda(), w.view(C0, 10).cuda(), dev(b), norm, dev(g) if norm != "none" else None,
                          dev(be) if norm != "none" else None, 1e-5, 5)
    assert torch.equal(got, again)  # fixed reduction order


def test_stem0_rejects_what_it_does_not_cover():
    from pytorch_models._hip import ops

    x = torch.zeros(1, 100, device="cuda")
    with pytest.raises(RuntimeError, match="pm_w2v_stem0"):
        ops.w2v_stem0(x, torch.zeros(8, 7, device="cuda"), None, "none", None, None, 0.0, 5)  # k != 10
    with pytest.raises(RuntimeError, match="pm_w2v_stem0"):
        ops.w2v_stem0(x, torch.zeros(1024, 10, device="cuda"), None, "none", None, None, 0.0, 5)  # C0 > 512
    with pytest.raises(ValueError, match="shorter"):
        ops.w2v_stem0(torch.zeros(1, 5, device="cuda"), torch.zeros(8, 10, device="cuda"), None, "none", None, None, 0.0, 5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("d,G,cgp,pl,pr", [(768, 16, 48, 64, 63), (64, 16, 8, 64, 63), (128, 16, 8, 9, 9), (96, 2, 48, 0, 3)])
def test_group_windows_is_an_exact_regrouping(dtype, d, G, cgp, pl, pr):
    from pytorch_models._hip import ops

    B, T = 3, 37
    x = synth_input("gw_x", (B, T, d), 6).to(dtype)
    cg = d // G
    want = torch.zeros(B, G, pl + T + pr, cgp, dtype=torch.bfloat16)
    want[:, :, pl : pl + T, :cg] = x.view(B, T, G, cg).permute(0, 2, 1, 3).to(torch.bfloat16)
    got = ops.group_windows(x.cuda(), G, cgp, pl, pr)
    assert torch.equal(got.cpu(), want)


def test_avgpool_time2():
    from pytorch_models._hip import ops

    for T in (2, 19, 64):
        x = synth_input("pool_x", (3, T, 128), 7).to(torch.bfloat16)
        want = x[:, : T // 2 * 2].float().view(3, T // 2, 2, 128).mean(2).to(torch.bfloat16)
        assert torch.equal(ops.avgpool_time2(x.cuda()).cpu(), want)


@pytest.mark.parametrize("affine,act,resid", [(True, "gelu", None), (False, "gelu", torch.bfloat16), (False, "none", torch.float32),
                                               (True, "none", torch.bfloat16)])
def test_layernorm_ex(affine, act, resid):
    from pytorch_models._hip import ops

    M, d = 67, 512
    x = synth_input("lnx_x", (M, d), 8, 2.0).to(torch.bfloat16)
    g, b = (synth_tensor("lnx.weight", (d,), 8), synth_tensor("lnx.bias", (d,), 8)) if affine else (None, None)
    r = None if resid is None else synth_input("lnx_r", (M, d), 8).to(resid)
    want = RA._norm_free(x.float(), 1, 1e-5)
    if affine:
        want = want * g + b
    want = (want if act == "none" else RT.activation(want, act)) + (0 if r is None else r.float())
    dev = lambda t: None if t is None else t.cuda()
    once_rounded(ops.layernorm(x.cuda(), dev(g), dev(b), 1e-5, act=act, resid=dev(r)), want)
    got32 = ops.layernorm(x.cuda(), dev(g), dev(b), 1e-5, torch.float32, act=act, resid=dev(r))
    torch.testing.assert_close(got32.cpu(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("C,Co,k,s,T", [(512, 512, 3, 2, 399), (512, 512, 2, 2, 49), (64, 128, 3, 2, 1279), (128, 128, 1, 1, 77)])
def test_conv_layer_as_strided_window_gemm(C, Co, k, s, T):
    """FeatureEncoder layers >= 1 (wav2vec2.py:32-38; SEW's k = 1 layers, sew.py:12-14) on pm_linear_bf16_ex."""
    from pytorch_models.audio.wav2vec2 import FeatureEncoder

    fe = FeatureEncoder((C, Co), (10, k), (5, s), bias=True, legacy=True)
    fill_module(fe, 9)
    bf16_round_(fe)
    B = 2
    h = synth_input("convgemm_x", (B, T, C), 9).to(torch.bfloat16)
    want = RT.activation(RA.conv1d_tl(fe[1][0].weight, fe[1][0].bias, h.float(), s), "gelu")
    fe = fe.cuda()
    w, b = fe._conv_weight(fe[1][0])
    from pytorch_models._hip import ops

    To = (T - k) // s + 1
    got = ops.linear_strided(h.cuda(), M=B * To, K=k * C, row_stride=s * C, rows_per_batch=To, batch_stride=T * C, w=w, bias=b,
                             act="gelu").view(B, To, Co)
    assert rel(got, want) < 1e-2


@pytest.mark.parametrize("d,k,pad,stride", [(768, 128, (64, 63), 1), (64, 128, (64, 63), 1), (128, 19, (9, 9), 1), (128, 31, (15, 14), 2),
                                            (1024, 128, (64, 63), 1), (512, 31, (15, 14), 2), (256, 19, (9, 9), 1),
                                            (160, 19, (9, 9), 1)])
def test_grouped_positional_conv(d, k, pad, stride):
    """wav2vec2.py:70-74 / data2vec_audio.py:25 / sew.py:24 on pm_grouped_conv_bf16 (48 / 4 / 8 / 64 / 32 / 16 channels per
    group; 10 channels per group falls back to one strided-window GEMM per group)."""
    from pytorch_models.audio import Wav2Vec2

    conv = torch.nn.Conv1d(d, d, k, stride=stride, groups=16)
    fill_module(conv, 10)
    bf16_round_(conv)
    B, T = (2, 51) if d != 768 else (3, 499)  # 499 steps: two 256-step tiles per clip, the second one ragged
    h = synth_input("pe_x", (B, T, d), 10).to(torch.bfloat16)
    To = (T + pad[0] + pad[1] - k) // stride + 1
    r = synth_input("pe_r", (B, To, d), 10).to(torch.bfloat16)
    want = RT.activation(RA.conv1d_tl(conv.weight, conv.bias, RA._pad_time(h.float(), *pad), stride, 16), "gelu") + r.float()
    got = Wav2Vec2.grouped_conv(conv.cuda(), h.cuda(), pad, "gelu", r.cuda())
    assert got.shape == (B, To, d) and rel(got, want) < 1e-2


def prep(m, seed):
    fill_module(m, seed)
    bf16_round_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.to(torch.bfloat16).cuda().eval(), sd


def model_cases():
    from pytorch_models.audio import SEW, Data2VecAudio, Wav2Vec2

    return dict(
        w2v_d64=(lambda: Wav2Vec2(2, 64), 82, lambda sd, x: RA.wav2vec2(sd, x), RA.STEM_STRIDES, False),
        w2v_legacy_post_d128=(lambda: Wav2Vec2(2, 128, stem_bias=False, stem_legacy=True, pre_norm=False), 83,
                              lambda sd, x: RA.wav2vec2(sd, x, pre_norm=False, legacy=True), RA.STEM_STRIDES, True),
        d2v_d128=(lambda: Data2VecAudio(2, 128), 84, RA.data2vec_audio, RA.STEM_STRIDES, False),
        sew_d128=(lambda: SEW(2, 128), 85, RA.sew, RA.SEW_STRIDES, True),
    )


@pytest.mark.parametrize("name", ["w2v_d64", "w2v_legacy_post_d128", "d2v_d128", "sew_d128"])
def test_models_end_to_end(golden, name):
    g = golden("audio_enc")
    make, seed, fwd, strides, legacy = model_cases()[name]
    m, sd = prep(make(), seed)
    x = synth_input("w2v_x", (2, 6400), 81)
    feat = m.feature_encoder.time_major(x.cuda())
    assert feat.dtype == torch.bfloat16
    assert rel(feat, RA.feature_encoder(sd, "feature_encoder.", x, strides, legacy)) < 2e-2
    assert rel(feat[..., ::8], g[name + "_feat_s8"]) < 3e-2
    assert rel(m.feature_encoder(x.unsqueeze(1).cuda()), RA.feature_encoder(sd, "feature_encoder.", x, strides, legacy).transpose(1, 2)) < 2e-2
    assert rel(m._features(x.cuda()), g[name + "_proj"]) < 3e-2
    y = m(x.cuda())
    assert y.dtype == torch.bfloat16 and y.shape == g[name].shape
    assert rel(y, fwd(sd, x)) < 2e-2, rel(y, fwd(sd, x))
    assert rel(y, g[name]) < 3e-2
    assert torch.equal(y, m(x.cuda()))
    if name == "sew_d128":
        assert y[:, -1].abs().max() == 0  # 19 frames: the zero frame appended after up-sampling
        assert rel(m(x[:, :6080].cuda()), g["sew_d128_even"]) < 3e-2


def test_fp32_model_and_batch_invariance(golden):
    """An fp32 instance is refused loudly (this family's stem / positional-conv kernels exist in bf16 only: it is not run
    through bf16 copies behind the caller's back); clips of a batch do not interact."""
    from pytorch_models.audio import Wav2Vec2

    g = golden("audio_enc")
    m = Wav2Vec2(2, 64)
    fill_module(m, 82)
    x = synth_input("w2v_x", (2, 6400), 81).cuda()
    with pytest.raises(NotImplementedError, match="bf16 parameters only"):
        m.cuda().eval()(x)
    m = m.to(torch.bfloat16).cuda().eval()
    y = m(x)
    assert y.dtype == torch.bfloat16 and rel(y, g["w2v_d64"]) < 3e-2
    assert torch.equal(m(x[1:]), y[1:])


def test_wav2vec2_base_geometry_10s_clip():
    """wav2vec2-base (12 x 768, legacy stem, post-norm) on 10 s of audio: 499 frames; the feature encoder's big GEMMs and
    the 48-channel positional-conv groups at their real sizes, against the oracle on the first clip."""
    from pytorch_models.audio import Wav2Vec2

    m, sd = prep(Wav2Vec2(2, 768, stem_bias=False, stem_legacy=True, pre_norm=False), 90)
    x = synth_input("w2v_10s", (2, 160000), 91)
    y = m(x.cuda())
    assert y.shape == (2, 499, 768)
    assert rel(y[:1], RA.wav2vec2(sd, x[:1], pre_norm=False, legacy=True)) < 2e-2


def test_graphed_forward_replays_the_eager_launches():
    """pytorch_models.graph.GraphedForward: one captured HIP graph gives bit-identical outputs to the eager forward, also on
    new input contents; shape changes are refused."""
    from pytorch_models.audio import Wav2Vec2
    from pytorch_models.graph import GraphedForward

    m, _ = prep(Wav2Vec2(2, 128, stem_bias=False, stem_legacy=True, pre_norm=False), 83)
    x1 = synth_input("w2v_x", (2, 6400), 81).cuda()
    x2 = synth_input("w2v_x2", (2, 6400), 82).cuda()
    g = GraphedForward(m, x1)
    assert torch.equal(g(x1), m(x1))
    assert torch.equal(g(x2), m(x2))
    assert torch.equal(g(x1), m(x1))
    with pytest.raises(ValueError, match="captured for"):
        g(x1[:, :3200])
    with pytest.raises(RuntimeError, match="HIP device"):
        GraphedForward(m, x1.cpu())


def test_no_projection_when_stem_width_equals_d_model_and_fp32_sew():
    """d_model = 512 = the stem's width: proj is the LayerNorm alone (wav2vec2.py:67-68), and the positional conv has 32
    channels per group.  SEW left in fp32 is refused; in bf16 it returns the stem's frame rate."""
    from pytorch_models.audio import SEW, Wav2Vec2

    m, sd = prep(Wav2Vec2(1, 512), 87)
    assert len(m.proj) == 1
    x = synth_input("w2v_x", (2, 6400), 81)
    y = m(x.cuda())
    assert y.shape == (2, 19, 512) and rel(y, RA.wav2vec2(sd, x)) < 2e-2
    s = SEW(2, 128)
    fill_module(s, 85)
    sd = {k: v.clone() for k, v in s.state_dict().items()}
    with pytest.raises(NotImplementedError, match="bf16 parameters only"):
        s.cuda().eval()(x.cuda())
    ys = s.to(torch.bfloat16).cuda().eval()(x.cuda())
    assert ys.dtype == torch.bfloat16 and ys.shape == (2, 19, 128) and rel(ys, RA.sew(sd, x)) < 3e-2
