"""GPU parity of the text models through the C ABI: GPT-2, GPT and BERT forwards against the
fp32 oracle on the same bf16-rounded weights and against the reference's own vectors (tests/golden/text.npz), and the
KV-cached greedy generation of GPT-2 against the oracle's full-recompute greedy loop.

bf16 tolerance as in test_hip_blocks.py: rel-L2 <= 2e-2 vs the oracle, 3e-2 vs the reference golden (fp32 weights)."""
import pytest
import torch

from oracle import ref_text as RX
from synthweights import bf16_round_, fill_module, synth_tokens

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rel(got, want):
    got, want = got.float().cpu(), want.float()
    return ((got - want).norm() / want.norm()).item()


def prep(m, seed):
    fill_module(m, seed)
    bf16_round_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.to(torch.bfloat16).cuda().eval(), sd


def test_gpt2_gpt_bert_forward(golden):
    from pytorch_models.text import BERT, GPT, GPT2

    g = golden("text")
    tok = synth_tokens("text_tok", (2, 16), 2000, 71)
    for cls, args, seed, fwd, name in ((GPT2, (2, 128), 72, RX.gpt2, "gpt2"), (GPT, (2, 128), 73, RX.gpt, "gpt")):
        m, sd = prep(cls(*args), seed)
        lg = m(tok.cuda())
        assert lg.dtype == torch.float32 and lg.shape == (2, 16, cls.vocab_size)
        assert rel(lg, fwd(sd, tok)) < 2e-2
        assert rel(lg[..., ::101], g[name + "_logits_s101"]) < 3e-2
        assert m(tok[0].cuda()).shape == (16, cls.vocab_size)  # unbatched, as the reference's generator calls it
    m, sd = prep(BERT(2000, 2, 128), 74)
    h = m(tok.cuda())
    assert h.shape == (2, 16, 128) and rel(h, RX.bert(sd, tok)) < 2e-2 and rel(h, g["bert_hidden"]) < 3e-2


def test_gpt2_kv_cached_greedy_matches_the_oracle_loop():
    """GPT2.generate (decode-step kernels, graph replay) vs the oracle's full-recompute greedy loop on the same
    bf16-rounded weights: identical ids, or a first difference only where the oracle's own top-2 margin is a near-tie."""
    from pytorch_models.text import GPT2, DecoderGenerator

    m, sd = prep(GPT2(2, 128), 72)
    tok = synth_tokens("text_tok", (2, 16), 2000, 71)
    P, n_new = 6, 24
    got = m.generate(tok[:, :P].cuda(), n_new).cpu()
    assert got.shape == (2, P + n_new) and torch.equal(got[:, :P], tok[:, :P])

    def kv_round(name, t):  # the cache holds bf16 k / v (the rounding point of the decode kernels)
        return t.to(torch.bfloat16).float() if name == "kv" else t

    want, margins = RX.greedy(RX.gpt2, sd, tok[:, :P], n_new, rp=kv_round)
    for b in range(2):
        diff = (got[b] != want[b]).nonzero()
        if len(diff):
            t = int(diff[0])
            assert float(margins[b, t - P]) < 2e-4, f"sequence {b}: ids differ at position {t} at a decisive margin"
    assert torch.equal(m.generate(tok[:, :P].cuda(), n_new, graph=False).cpu(), got)  # eager == graph replay
    # the reference-shaped front end: tokenizer in, text out; greedy goes through the same KV-cached path
    class Tok:
        eos_token_id = None

        def encode(self, s):
            return [int(t) for t in s.split()]

        def decode(self, ids):
            return " ".join(str(int(i)) for i in ids)

    text = DecoderGenerator(m, Tok()).generate(Tok().decode(tok[0, :P]), max_tokens=n_new)
    assert text == Tok().decode(got[0])


@pytest.mark.parametrize("B", [1, 17, 64])
def test_gpt2_greedy_batch_sizes(B):
    """One sequence, a ragged 17 and the 64-sequence maximum (1, 2 and 4 row tiles of the decode projections): every
    sequence decodes exactly as it does alone in a batch of one (batch invariance of the KV-cached path)."""
    from pytorch_models.text import GPT2

    m, _ = prep(GPT2(2, 128), 72)
    tok = synth_tokens("text_tok_many", (64, 8), 2000, 80)[:B]
    got = m.generate(tok.cuda(), 12).cpu()
    assert got.shape == (B, 20)
    for b in sorted({0, B // 2, B - 1}):
        assert torch.equal(m.generate(tok[b:b + 1].cuda(), 12).cpu()[0], got[b]), b
    with pytest.raises(NotImplementedError, match="64 sequences"):
        m.generate(synth_tokens("text_tok_65", (65, 4), 2000, 81).cuda(), 2)


def test_topk_sampling_kernel_distribution():
    """pm_dec_sample_topk on synthetic logits: k = 1 is the arg-max (lowest index on ties); for k = 5 every draw is one of
    the five largest, the same seed repeats, and over 4096 (seed, position, sequence) keys the empirical frequencies
    follow softmax(top-5 logits) (total variation < 0.05)."""
    from pytorch_models._hip import check, lib

    B, V, d, k = 32, 1000, 64, 5
    gen = torch.Generator().manual_seed(3)
    logits = torch.randn(B, V, generator=gen) * 2
    logits[:, 7] = logits[:, 3]  # exact ties
    dev = dict(device="cuda")
    lg = logits.cuda()
    E = torch.zeros(V, d, dtype=torch.bfloat16, **dev)
    pos_tab = torch.zeros(8, d, dtype=torch.float32, **dev)
    x = torch.empty(B, d, dtype=torch.float32, **dev)
    prompt = torch.zeros(B, 1, dtype=torch.int64, **dev)
    tok_cur = torch.zeros(B, dtype=torch.int64, **dev)
    tokens = torch.zeros(B, 4, dtype=torch.int64, **dev)
    pos = torch.zeros(1, dtype=torch.int32, **dev)
    ticket = torch.zeros(1, dtype=torch.int32, **dev)

    def draw(kk, seed):
        pos.zero_()
        check(lib().pm_dec_sample_topk(lg.data_ptr(), V, V, kk, seed, pos.data_ptr(), prompt.data_ptr(), 1, tok_cur.data_ptr(),
                                       tokens.data_ptr(), 4, E.data_ptr(), pos_tab.data_ptr(), x.data_ptr(), d, ticket.data_ptr(),
                                       B, None), "pm_dec_sample_topk")
        assert int(pos) == 1 and int(ticket) == 0
        return tok_cur.cpu().clone()

    assert torch.equal(draw(1, 0), logits.argmax(-1))
    top = logits.topk(k, -1)
    counts = torch.zeros(B, k)
    first = draw(k, 11)
    assert torch.equal(draw(k, 11), first)
    for seed in range(128):
        t = draw(k, seed)
        hit = t[:, None] == top.indices
        assert hit.any(1).all()
        counts += hit.float()
    want = top.values.softmax(-1)
    tv = 0.5 * (counts / 128 - want).abs().sum(1)
    assert float(counts.sum()) == B * 128 and float(tv.mean()) < 0.08, tv


def test_gpt2_topk_sampling_stays_inside_the_oracle_top_k():
    """GPT2.generate(topk = 4): repeatable for a seed, different seeds differ, and every sampled token is one of the four
    most likely continuations of its prefix according to the fp32 oracle (near-ties at the boundary excepted)."""
    from pytorch_models.text import GPT2

    m, sd = prep(GPT2(2, 128), 72)
    tok = synth_tokens("text_tok", (2, 16), 2000, 71)
    P, n_new, k = 6, 16, 4
    a = m.generate(tok[:, :P].cuda(), n_new, topk=k, seed=5).cpu()
    assert torch.equal(m.generate(tok[:, :P].cuda(), n_new, topk=k, seed=5).cpu(), a)
    assert not torch.equal(m.generate(tok[:, :P].cuda(), n_new, topk=k, seed=6).cpu(), a)
    lg = RX.gpt2(sd, a[:, :-1])  # oracle logits for every prefix of the sampled text
    for b in range(2):
        for t in range(P, P + n_new):
            row = lg[b, t - 1]
            kth = row.topk(k).values[-1]
            assert row[a[b, t]] >= kth - 2e-3, (b, t)


def test_gpt2_size_geometry_runs_one_step():
    """GPT-2 small geometry (12 x 768, 12 heads), batch 4: logits shape / finiteness and 8 greedy tokens in range."""
    from pytorch_models.text import GPT2

    m, _ = prep(GPT2.from_hf("gpt2"), 78)
    tok = synth_tokens("text_tok_b", (4, 32), 50257, 79)
    lg = m(tok.cuda())
    assert lg.shape == (4, 32, 50257) and torch.isfinite(lg).all()
    ids = m.generate(tok[:, :8].cuda(), 8)
    assert ids.shape == (4, 16) and int(ids.min()) >= 0 and int(ids.max()) < 50257
