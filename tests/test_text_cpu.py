"""CPU: the text models of SURVEY.md 8(f) rows 2-3 - the oracle (oracle/ref_text.py) against vectors captured from the
reference (tests/golden/text.npz, made by tests/golden/make_golden.py text), constructor contracts, and the HF / OpenAI
weight loaders against digests of what the reference's loaders produce (tests/golden/text_converters.json)."""
import json
import os

import numpy as np
import pytest
import torch

import ckpt_synth as C
from oracle import ref_text as RX
from synthweights import fill_module, synth_tokens

torch.set_grad_enabled(False)
TOL = dict(rtol=2e-5, atol=2e-5)
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_converters.json")))


def sd_of(m, seed):
    fill_module(m, seed)
    return {k: v.clone() for k, v in m.state_dict().items()}


def digest(t):
    f = t.double().flatten()
    w = 1.0 + (torch.arange(f.numel(), dtype=torch.float64) % 251) / 251.0
    return torch.tensor([f.sum().item(), f.abs().sum().item(), (f * w).sum().item()], dtype=torch.float64)


def check_logits(lg, g, name):
    torch.testing.assert_close(lg[..., ::101], g[name + "_logits_s101"], **TOL)
    assert torch.equal(lg.argmax(-1), g[name + "_argmax"])
    got, want = digest(lg), g[name + "_digest"]
    assert ((got - want).abs() <= 1e-5 * want[1].abs()).all(), (got, want)


def test_oracle_gpt2_gpt_bert_match_the_reference(golden):
    from pytorch_models.text import BERT, GPT, GPT2

    g = golden("text")
    tok = synth_tokens("text_tok", (2, 16), 2000, 71)
    check_logits(RX.gpt2(sd_of(GPT2(2, 128), 72), tok), g, "gpt2")
    check_logits(RX.gpt(sd_of(GPT(2, 128), 73), tok), g, "gpt")
    torch.testing.assert_close(RX.bert(sd_of(BERT(2000, 2, 128), 74), tok), g["bert_hidden"], **TOL)


def test_oracle_greedy_matches_the_reference_generator(golden):
    """text/generator.py:23-35 with topk=1: 12 new tokens from a 6-token prompt, per sequence."""
    from pytorch_models.text import GPT2

    g = golden("text")
    tok = synth_tokens("text_tok", (2, 16), 2000, 71)
    ids, _ = RX.greedy(RX.gpt2, sd_of(GPT2(2, 128), 72), tok[:, :6], 12)
    assert torch.equal(ids, g["gpt2_greedy"])


def test_constructors():
    from pytorch_models.text import BERT, GPT, GPT2

    for tag, (n, d) in {"gpt2": (12, 768), "gpt2-medium": (24, 1024)}.items():
        m = GPT2.from_hf(tag)
        assert len(m.layers) == n and m.pos_embs.shape == (1024, d) and m.token_embs.weight.shape == (50257, d)
        assert all(l.pre_norm and l.ca is None and l.mlp.act_name == "approximate_gelu" for l in m.layers)
    with pytest.raises(KeyError):
        GPT2.from_hf("gpt3")
    with pytest.raises(NotImplementedError, match="network"):
        GPT2.from_hf("gpt2", pretrained=True)
    m = GPT(2, 128)
    assert m.token_embs.weight.shape == (40478, 128) and m.pos_embs.shape == (512, 128) and not hasattr(m, "norm")
    assert all(not l.pre_norm for l in m.layers)
    b = BERT(30522, 2, 128)
    assert b.token_embs.weight.shape[0] == 30528 and b.norm.eps == 1e-12 and all(not l.pre_norm for l in b.layers)  # vocab padded to 64
    r = BERT.from_config(dict(model_type="roberta", vocab_size=1000, num_hidden_layers=2, hidden_size=128,
                              max_position_embeddings=66, layer_norm_eps=1e-5))
    assert r.pos_embs.shape[0] == 64 and r.norm.eps == 1e-5
    with pytest.raises(NotImplementedError, match="network"):
        BERT.from_hf("roberta-base")
    # same state_dict keys as the reference classes (the names are contract)
    assert sorted(GPT2(1, 64).state_dict()) == sorted(
        ["token_embs.weight", "pos_embs", "norm.weight", "norm.bias"] + [f"layers.0.{k}" for k in (
            "sa_norm.weight", "sa_norm.bias", "sa.q_proj.weight", "sa.q_proj.bias", "sa.k_proj.weight", "sa.k_proj.bias",
            "sa.v_proj.weight", "sa.v_proj.bias", "sa.out_proj.weight", "sa.out_proj.bias", "mlp_norm.weight", "mlp_norm.bias",
            "mlp.linear1.weight", "mlp.linear1.bias", "mlp.linear2.weight", "mlp.linear2.bias")])


def check(model, name):
    got, want = C.state_digest(model.state_dict()), GOLD[name]
    assert sorted(got) == sorted(want)
    for k in want:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-9, atol=1e-9, err_msg=f"{name}: {k}")


def test_hf_loaders_place_every_parameter_like_the_reference():
    from pytorch_models.text import BERT, GPT, GPT2

    m = GPT2(2, 128)
    m.load_hf_state_dict(C.hf_gpt2(2, 128, GPT2.vocab_size, GPT2.max_seq_len, seed=75))
    check(m, "hf_gpt2")
    for name, rob in (("hf_bert", False), ("hf_roberta", True)):
        b = BERT(1024, 2, 128, max_seq_len=64)
        b.load_hf_state_dict(C.hf_bert(2, 128, 1024, 64, roberta=rob, seed=76))
        check(b, name)
    # GPT: the OpenAI parameter list - q | k | v chunks of c_attn land transposed in the three projections
    ps = C.openai_gpt_params(2, 128, 500, GPT.max_seq_len, seed=77)
    gm = GPT(2, 128)
    gm.load_openai_params(ps)
    torch.testing.assert_close(gm.token_embs.weight[:500], ps[1])
    torch.testing.assert_close(gm.layers[1].sa.k_proj.weight, ps[2 + 12][0][:, 128:256].T)
    torch.testing.assert_close(gm.layers[1].mlp.linear2.weight, ps[2 + 12 + 8][0].T)
    torch.testing.assert_close(gm.layers[0].mlp_norm.bias, ps[2 + 11])


def test_generator_token_level_api_on_a_stub_model():
    """DecoderGenerator falls back to the full-forward loop for models without a KV-cached generate (here: a stub)."""
    from pytorch_models.text import DecoderGenerator

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.layers = []

        def forward(self, toks):  # next token = (last + 1) mod 7, as one-hot logits
            out = torch.zeros(len(toks), 7)
            out[torch.arange(len(toks)), (toks + 1) % 7] = 1.0
            return out

    class Tok:
        eos_token_id = 5

        def encode(self, s):
            return [int(t) for t in s.split()]

        def decode(self, ids):
            return " ".join(str(i) for i in ids)

    gen = DecoderGenerator(Stub(), Tok())
    assert gen.generate("1 2", max_tokens=10) == "1 2 3 4 5"  # stops after eos (kept)
    assert gen.generate_ids([6], max_tokens=3) == [6, 0, 1, 2]
    assert len(gen.generate_ids([0], max_tokens=4, topk=3, generator=torch.Generator().manual_seed(0))) == 5
