"""GPU parity of T5 v1.1 through the C ABI: pm_rmsnorm / pm_geglu / pm_embed_tokens without positions against the fp32
oracle (oracle/ref_t5.py), then T5Encoder / T5Decoder / T5Model end to end against the oracle on the same bf16-rounded
weights and against the reference's own vectors (tests/golden/t5.npz), and the greedy loop.

Tolerances as in test_hip_blocks.py: kernels that round once to bf16 |err| <= 4e-3 |want| + 1e-3; whole stacks rel-L2
<= 2e-2 vs the oracle, 3e-2 vs the reference golden (fp32 weights)."""
import pytest
import torch

from oracle import ref_t5 as R5
from oracle import ref_transformer as RT
from synthweights import bf16_round_, fill_module, synth_input, synth_tensor, synth_tokens

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DIM, HEADS, LAYERS, MLP = 512, 6, 2, 1024


def rel(got, want):
    got, want = got.float().cpu(), want.float()
    return ((got - want).norm() / want.norm()).item()


def prep(m, seed):
    fill_module(m, seed)
    bf16_round_(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.to(torch.bfloat16).cuda().eval(), sd


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rmsnorm(dtype):
    from pytorch_models._hip import ops

    x = synth_input("rms_x", (67, 512), 3, 2.0).to(dtype)
    g = synth_tensor("rms.weight", (512,), 3)
    want = R5.rmsnorm(g, x.float())
    torch.testing.assert_close(ops.rmsnorm(x.cuda(), g.cuda(), 1e-5, torch.float32).cpu(), want, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(ops.rmsnorm(x.cuda(), g.cuda(), 1e-5, torch.bfloat16).float().cpu(), want, rtol=4e-3, atol=1e-3)
    # a constant offset must NOT be removed (this is what distinguishes it from LayerNorm)
    shifted = ops.rmsnorm((x.float() + 3.0).cuda(), g.cuda(), 1e-5, torch.float32).cpu()
    torch.testing.assert_close(shifted, R5.rmsnorm(g, x.float() + 3.0), rtol=2e-5, atol=2e-5)


def test_geglu_and_embedding_without_positions():
    from pytorch_models._hip import ops

    h = synth_input("geglu_h", (37, 2 * 1024), 4, 1.5).to(torch.bfloat16)
    want = RT.activation(h[:, :1024].float(), "approximate_gelu") * h[:, 1024:].float()
    torch.testing.assert_close(ops.geglu(h.cuda()).float().cpu(), want, rtol=4e-3, atol=1e-3)
    E = synth_tensor("emb.weight", (100, 64), 4).to(torch.bfloat16)
    tok = synth_tokens("emb_tok", (3, 7), 100, 4)
    assert torch.equal(ops.embed_tokens(tok.cuda(), E.cuda(), None).cpu(), E[tok])


def test_encoder_decoder_model(golden):
    from pytorch_models.text import T5Decoder, T5Encoder, T5Model

    g = golden("t5")
    x, mem = synth_input("t5_x", (2, 64, DIM), 91), synth_input("t5_mem", (2, 32, DIM), 91)
    xb, memb = x.to(torch.bfloat16), mem.to(torch.bfloat16)
    m, sd = prep(T5Encoder(DIM, HEADS, LAYERS, MLP), 92)
    y = m(xb.cuda())
    assert y.dtype == torch.bfloat16 and rel(y, R5.encoder(sd, "", xb.float())) < 2e-2 and rel(y[..., ::4], g["encoder"]) < 3e-2
    assert rel(m(xb[0].cuda())[..., ::4], g["encoder_unbatched"]) < 3e-2  # unbatched (L, d), tests/text/test_t5.py:27-29
    m, sd = prep(T5Decoder(DIM, HEADS, LAYERS, MLP), 93)
    y = m(xb.cuda(), memb.cuda())
    assert rel(y, R5.decoder(sd, "", xb.float(), memb.float())) < 2e-2 and rel(y[..., ::4], g["decoder"]) < 3e-2
    # causality: a later target position cannot change an earlier output
    x2 = xb.clone()
    x2[:, 40:] = 0
    assert torch.equal(m(x2.cuda(), memb.cuda())[:, :40], y[:, :40])
    m, sd = prep(T5Model(2000, DIM, HEADS, LAYERS, MLP), 94)
    tok, tgt = synth_tokens("t5_tok", (2, 64), 1000, 95), synth_tokens("t5_tgt", (2, 32), 1000, 95)
    lg = m(tok.cuda(), tgt.cuda())
    assert lg.dtype == torch.float32 and lg.shape == (2, 32, 2000)
    assert rel(lg, R5.model(sd, tok, tgt)) < 2e-2 and rel(lg[..., ::7], g["model_logits_s7"]) < 3e-2
    assert torch.equal(lg, m(tok.cuda(), tgt.cuda()))
    assert m(tok[0].cuda(), tgt[0].cuda()).shape == (32, 2000)


def test_greedy_ids_and_fp32_model(golden):
    """generate_ids (the reference generator's loop on token ids) against the oracle's loop on the same bf16-rounded
    weights: identical ids, or a first difference only where the oracle's own top-2 margin is a near-tie; an fp32 model
    (the reference's default) is refused (T5's own kernels are bf16); cast to bf16 it gives fp32 logits."""
    from pytorch_models.text import T5Model

    g = golden("t5")
    m, sd = prep(T5Model(2000, DIM, HEADS, LAYERS, MLP), 94)
    tok = synth_tokens("t5_tok", (2, 64), 1000, 95)
    want, margins = R5.greedy(sd, tok[0], 12)
    got = m.generate_ids(tok[0].cuda(), 12).cpu()
    n = min(len(got), len(want))
    diff = (got[:n] != want[:n]).nonzero()
    if len(diff) or len(got) != len(want):
        first = int(diff[0]) if len(diff) else n
        assert margins[first - 1] < 0.05, (got, want, margins)
    m32 = T5Model(2000, DIM, HEADS, LAYERS, MLP)
    fill_module(m32, 94)
    m32 = m32.cuda().eval()
    tgt = synth_tokens("t5_tgt", (2, 32), 1000, 95)
    with pytest.raises(NotImplementedError, match="bf16 parameters only"):  # T5's own kernels are bf16: refused, not down-cast
        m32(tok.cuda(), tgt.cuda())
    lg = m32.to(torch.bfloat16)(tok.cuda(), tgt.cuda())
    assert lg.dtype == torch.float32 and rel(lg[..., ::7], g["model_logits_s7"]) < 3e-2
