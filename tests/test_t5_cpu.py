"""CPU: T5 v1.1 - the oracle (oracle/ref_t5.py) against vectors captured from the reference (tests/golden/t5.npz, made by
tests/golden/make_golden.py t5), the relative-position bucket tables of this build's host code, constructor contracts and
the t5x checkpoint conversion against a digest of the reference's own conversion (tests/golden/t5_converter.json)."""
import json
import os

import pytest
import torch

import ckpt_synth as C
from oracle import ref_t5 as R5
from synthweights import fill_module, synth_input, synth_tokens

torch.set_grad_enabled(False)
TOL = dict(rtol=2e-5, atol=1e-4)
DIM, HEADS, LAYERS, MLP = 512, 6, 2, 1024


def sd_of(m, seed):
    fill_module(m, seed)
    return {k: v.clone() for k, v in m.state_dict().items()}


def digest(t):
    f = t.double().flatten()
    w = 1.0 + (torch.arange(f.numel(), dtype=torch.float64) % 251) / 251.0
    return torch.tensor([f.sum().item(), f.abs().sum().item(), (f * w).sum().item()], dtype=torch.float64)


def test_oracle_matches_the_reference(golden):
    from pytorch_models.text import T5Decoder, T5Encoder, T5Model

    g = golden("t5")
    x, mem = synth_input("t5_x", (2, 64, DIM), 91), synth_input("t5_mem", (2, 32, DIM), 91)
    sd = sd_of(T5Encoder(DIM, HEADS, LAYERS, MLP), 92)
    torch.testing.assert_close(R5.encoder(sd, "", x)[..., ::4], g["encoder"], **TOL)
    torch.testing.assert_close(R5.encoder(sd, "", x[0])[..., ::4], g["encoder_unbatched"], **TOL)
    sd = sd_of(T5Decoder(DIM, HEADS, LAYERS, MLP), 93)
    torch.testing.assert_close(R5.decoder(sd, "", x, mem)[..., ::4], g["decoder"], **TOL)
    sd = sd_of(T5Model(2000, DIM, HEADS, LAYERS, MLP), 94)
    tok, tgt = synth_tokens("t5_tok", (2, 64), 1000, 95), synth_tokens("t5_tgt", (2, 32), 1000, 95)
    lg = R5.model(sd, tok, tgt)
    torch.testing.assert_close(lg[..., ::7], g["model_logits_s7"], **TOL)
    assert torch.equal(lg.argmax(-1), g["model_argmax"])
    got, want = digest(lg), g["model_digest"]
    assert ((got - want).abs() <= 1e-5 * want[1].abs()).all(), (got, want)
    ids, _ = R5.greedy(sd, tok[0], 12)
    assert torch.equal(ids, g["greedy"])


@pytest.mark.parametrize("L", [8, 64, 200])
def test_bucket_tables_match_the_reference(golden, L):
    """Host index arithmetic of RelativePositionBias (this build) and of the oracle, against the reference's tables."""
    from pytorch_models.text.t5 import RelativePositionBias

    g = golden("t5")
    rp = RelativePositionBias(6)
    for bi, key in ((True, f"buckets_bi_{L}"), (False, f"buckets_uni_{L}")):
        want = g[key].long()
        assert torch.equal(rp.buckets(L, bi), want)
        assert torch.equal(R5.buckets(L, bi), want)
    assert int(g[f"buckets_bi_{L}"].max()) <= 31 and int(g[f"buckets_uni_{L}"].max()) <= 31


def test_constructors_and_converter():
    from pytorch_models.text import T5Model
    from pytorch_models.text.t5 import GEGLU, LayerNorm

    m = T5Model.from_t5x("t5_1_1-small")
    assert m.token_embs.weight.shape == (32128, 512) and len(m.encoder.layers) == 8 and m.encoder.layers[0].sa.n_heads == 6
    assert m.encoder.layers[0].sa.q_proj.weight.shape == (384, 512) and m.encoder.layers[0].sa.q_proj.bias is None
    assert isinstance(m.decoder.layers[0].mlp[0], GEGLU) and m.decoder.layers[0].ca is not None and m.encoder.layers[0].ca is None
    assert isinstance(m.encoder.norm, LayerNorm) and m.encoder.norm.weight.abs().sum() == 0  # zeros, like the reference (t5.py:18)
    assert T5Model.from_t5x("mt5-small").token_embs.weight.shape[0] == 250112
    with pytest.raises(NotImplementedError, match="no\\s+network"):
        T5Model.from_t5x("flan_t5-small", pretrained=True)
    with pytest.raises(RuntimeError, match="HIP devices only"):
        m.encoder.norm(torch.zeros(1, 512))
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "t5_converter.json")))
    m = T5Model(500, 128, 2, 2, 256)
    assert sorted(m.state_dict()) == sorted(gold)  # same parameter names as the reference
    m.load_t5x_checkpoint(C.t5x_flat(2, 128, 2, 256, 500, seed=96))
    got = C.state_digest(m.state_dict())
    for k, want in gold.items():
        w, gt = torch.tensor(want, dtype=torch.float64), torch.tensor(got[k], dtype=torch.float64)
        assert ((w - gt).abs() <= 1e-6 * w[1].abs() + 1e-9).all(), (k, want, got[k])
