"""The persistent decode-step kernel (csrc/decode_persist.hip, pm_dec_layers: every layer of a step in ONE launch, stages
handed to each other through agent-scope arrival counters) against the oracle and against the launch-per-stage path.

Same gate as tests/test_hip_decode.py: token ids bit-exact vs oracle.ref_whisper.greedy_cached on the same memory and
storage points; plus: hidden state after a step equal to the launch path's to fp32 rounding, graph replay == eager ==
second run bit for bit (the hand-offs add no order dependence), no hand-off timed out (err word), and the geometry
corners of the work distribution: ragged 16-row tiles, several row tiles, heads that do not divide the workgroup count,
decoder-only stacks (no cross block), d_model 384 / 512 / 768 / 1280."""
import os

import pytest
import torch

from oracle import ref_whisper as RW
from synthweights import bf16_round_, fill_module, synth_input, synth_tokens

from pytorch_models import _hip

# an EXPERIMENT (csrc/experiments/decode_persist.hip): runs against build/libpm_mi355x_exp.so (`make experiments`,
# PM_MI355X_LIB=<that file>); with the product library these tests are skipped
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(_hip.LIB_PATH) or not _hip.has_experiments(),
                                                  reason="experiments build only (make experiments; PM_MI355X_LIB)")]
torch.set_grad_enabled(False)


def kv_round(name, t):
    return t.to(torch.bfloat16).float() if name == "kv" else t


def _whisper(vocab, n_layers, d, seed):
    from pytorch_models.audio2text import Whisper

    w = Whisper(vocab, n_layers, d).eval()
    fill_module(w, seed)
    bf16_round_(w)
    sd = {k: v.clone() for k, v in w.state_dict().items()}
    return w.to(torch.bfloat16).cuda(), sd


def _decoders(w, memory, prompt, n_new, **kw):
    from pytorch_models.audio2text.generate import GreedyDecoder

    return (GreedyDecoder(w.decoder, memory, prompt.cuda(), n_new, path="persistent", **kw),
            GreedyDecoder(w.decoder, memory, prompt.cuda(), n_new, path="launches", **kw))


def _first_mismatch_ok(toks, want, margins, P):
    toks = toks.cpu()
    for b in range(toks.shape[0]):
        diff = (toks[b] != want[b]).nonzero()
        if len(diff):
            t = int(diff[0])
            m = float(margins[b, t - P])
            print(f"sequence {b}: first mismatch at position {t}: margin {m:.3e}")
            assert m < 2e-4, "token id mismatch at a decisive margin"


@pytest.mark.parametrize("d,n_layers,B,S", [(128, 2, 2, 50), (384, 2, 3, 200), (512, 3, 17, 333), (256, 1, 33, 64), (1280, 1, 2, 96)])
def test_hidden_state_and_ids_match_the_launch_path_and_the_oracle(d, n_layers, B, S):
    """One geometry per corner: the hidden state after EVERY step equals the launch path's up to what the cached K / V's
    bf16 rounding makes of fp32 summation-order noise (the two paths order the attention block's LayerNorm sums differently;
    a last-bit difference in k or v can flip its bf16 rounding: ~4e-3 of one element, ~1e-4 in the hidden state - a wrong
    hand-off would show as O(0.1 .. 1)), ids equal the oracle's."""
    w, sd = _whisper(1000, n_layers, d, 70 + d % 7)
    memory = synth_input(f"ps_mem{d}", (B, S, d), 71).to(torch.bfloat16).cuda()
    prompt = synth_tokens(f"ps_p{d}", (B, 3), 1000, 72)
    n_new = 9
    ps, ln = _decoders(w, memory, prompt, n_new, margins=True)
    assert ps.path == "persistent" and ln.path == "launches" and len(ps.launches) < len(ln.launches)
    ps.reset()
    ln.reset()
    st = torch.cuda.current_stream().cuda_stream

    def layers_then_tail(dec):  # everything but the last launch (token choice + next embedding row, which overwrites x)
        for fn, args in dec.launches[:-1]:
            assert fn(*args[:-1], st) == 0
        hidden = dec.x.clone()
        fn, args = dec.launches[-1]
        assert fn(*args[:-1], st) == 0
        return hidden

    for _ in range(ps.n_steps):
        torch.testing.assert_close(layers_then_tail(ps), layers_then_tail(ln), rtol=2e-3, atol=2e-3)
    ps.check()
    assert int(ps.err.item()) == 0
    want, margins = RW.greedy_cached(sd, "decoder.", prompt, memory.float().cpu(), n_new, rp=kv_round)
    _first_mismatch_ok(ps.tokens, want, margins, 3)
    assert torch.equal(ps.tokens, ln.tokens)
    # graph replay == eager == a second run
    t1 = ps.run(graph=True).clone()
    t2 = ps.run(graph=True).clone()
    assert torch.equal(t1, ps.tokens) and torch.equal(t1, t2)
    ps.check()


@pytest.mark.parametrize("tag,seed", [("tiny", 55), ("base", 56)])
def test_greedy_ids_bit_exact_vs_oracle_persistent(tag, seed):
    from pytorch_models.audio2text import Whisper, WhisperPreprocessor

    w = Whisper.from_openai(tag).eval()
    fill_module(w, seed)
    bf16_round_(w)
    sd = {k: v.clone() for k, v in w.state_dict().items()}
    w = w.to(torch.bfloat16).cuda()
    wave = synth_input(f"w_wave_{tag}", (2, 480000), seed, scale=0.1)
    memory = w.encoder(WhisperPreprocessor(tag).cuda()(wave.cuda()))
    prompt = synth_tokens(f"w_prompt_{tag}", (2, 4), 51865, seed)
    toks = w.decoder.generate(memory, prompt.cuda(), 32, path="persistent")
    want, margins = RW.greedy_cached(sd, "decoder.", prompt, memory.float().cpu(), 32, rp=kv_round)
    assert torch.equal(toks.cpu(), want)
    assert torch.equal(w.decoder.generate(memory, prompt.cuda(), 32, path="launches"), toks)
    assert torch.equal(w.decoder.generate(memory, prompt.cuda(), 32, path="persistent", graph=False), toks)


def test_batch32_full_length_under_load():
    """BASELINE configs[2] decode geometry (32 sequences, 8 layers, 1500 memory rows, 227 steps = 227 x 48 hand-off seams with
    every CU streaming).  (a) the hidden state after each of the first steps stays within 5e-2 of the launch path's for
    every sequence - the two forms order the attention block's LayerNorm sums differently and the bf16 rounding of the
    cached k / v turns a last-bit difference into ~1e-2 of a residual stream of magnitude ~10 (measured 1.2e-2; a wrong or
    stale hand-off shows as O(1 .. 10)); (b) the full-length run is bit-identical when repeated (no order dependence in
    the hand-offs) and no hand-off timed out; (c) ids: pinned to the oracle by the batch-2 test above - with random weights
    the logits are flat, so against the launch path only the agreement is reported."""
    from pytorch_models.audio2text import Whisper

    w = Whisper.from_openai("base").eval()
    fill_module(w, 56)
    w = w.to(torch.bfloat16).cuda()
    memory = synth_input("ps_mem_b32", (32, 1500, 512), 5).to(torch.bfloat16).cuda()
    prompt = synth_tokens("ps_p_b32", (32, 4), 51865, 5)
    ps, ln = _decoders(w, memory, prompt, 224)
    ps.reset()
    ln.reset()
    st = torch.cuda.current_stream().cuda_stream

    def layers_then_tail(dec):
        for fn, args in dec.launches[:-1]:
            assert fn(*args[:-1], st) == 0
        hidden = dec.x.clone()
        fn, args = dec.launches[-1]
        assert fn(*args[:-1], st) == 0
        return hidden

    for _ in range(6):
        a, b = layers_then_tail(ps), layers_then_tail(ln)
        assert float((a - b).abs().max()) < 5e-2 and float(b.abs().max()) > 1.0
    a = ps.run().clone()
    ps.check()
    b = ln.run().clone()
    print(f"ids equal to the launch path's: {(a == b).float().mean().item():.3f} of {a.numel()}")
    assert int(a.min()) >= 0 and int(a.max()) < 51865 and torch.equal(a[:, :4].cpu(), prompt)
    assert torch.equal(ps.run(), a)
    ps.check()
    assert int(ps.err.item()) == 0


def test_decoder_only_stack_gpt2_geometry():
    """No cross block, d_model 768 (12 heads: 32 x 12 tasks on 256 workgroups), tanh-GELU, top-k sampling behind it."""
    from pytorch_models.text import GPT2

    m = GPT2(3, 768).eval()
    fill_module(m, 91)
    m = m.to(torch.bfloat16).cuda()
    prompt = synth_tokens("ps_gpt2", (32, 5), 50257, 91).cuda()
    a = m.generate(prompt, 40, path="persistent")
    b = m.generate(prompt, 40, path="launches")
    assert (a == b).float().mean() > 0.9  # near-ties aside (see the batch-32 test)
    assert torch.equal(a, m.generate(prompt, 40, path="persistent"))
    s1 = m.generate(prompt, 20, topk=8, seed=3, path="persistent")
    assert torch.equal(s1, m.generate(prompt, 20, topk=8, seed=3, path="persistent"))
