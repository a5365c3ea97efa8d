"""GPU: Whisper's decoding-time logit filters (pm_dec_whisper_rules) against oracle/ref_whisper_rules.py - PARITY UNPINNED: the
reference has no Whisper decoding and OpenAI's package is not in the image, so the oracle restates the published rules and this
file additionally checks invariants of generated token streams (first token a timestamp, timestamps in non-decreasing pairs)."""
import random

import pytest
import torch

from oracle import ref_whisper_rules as RR
from synthweights import bf16_round_, fill_module, synth_input, synth_tokens

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
V, EOT, TB, NOTS = 700, 500, 560, 555  # text ids [0, 500), specials [500, 560), timestamps [560, 700)


def _history_cases():
    ts = lambda k: TB + k
    return [
        [],                                   # first token: must be a timestamp, blank / eot suppressed, initial cap
        [ts(0)],                              # lone opening stamp: text must follow; stamps masked
        [ts(3), 17],                          # open segment with text: anything but earlier stamps
        [ts(3), 17, 42, ts(9)],               # text then a stamp: the segment must be closed -> no text below eot
        [ts(3), 17, ts(9), ts(9)],            # closed pair: text must follow
        [ts(3), 17, ts(9), ts(9), 5, 6],      # later segment: stamps below or equal 9 masked
        [ts(0), ts(0)],                       # empty first segment
        [ts(120), 1, 2, 3],                   # late stamp
        [3, 4, 5],                            # no stamp yet (rules entered mid-way)
    ]


@pytest.mark.parametrize("hist", _history_cases(), ids=lambda h: "h" + "_".join(str(x) for x in h) if h else "first")
@pytest.mark.parametrize("favour_ts", [False, True])
def test_rules_kernel_matches_the_restated_rules(hist, favour_ts):
    from pytorch_models._hip import ops

    P, B = 3, 4
    rng = random.Random(len(hist) * 7 + favour_ts)
    g = torch.Generator().manual_seed(len(hist) * 13 + favour_ts)
    logits = torch.randn(B, V, generator=g) * 3.0
    if favour_ts:
        logits[:, TB:] += 4.0  # the probability rule fires: timestamps outweigh every text token
    rules = RR.Rules(eot=EOT, timestamp_begin=TB, no_timestamps=NOTS, max_initial_timestamp=50,
                     suppress=sorted(rng.sample(range(0, TB), 20)) + [TB + 5], blank=[11, EOT])
    tokens = torch.zeros(B, P + 16, dtype=torch.int64)
    tokens[:, :P] = torch.tensor([501, 502, 503])
    for b in range(B):
        tokens[b, P : P + len(hist)] = torch.tensor(hist, dtype=torch.int64) if hist else tokens[b, P:P]
    pos = torch.tensor([P + len(hist) - 1], dtype=torch.int32)
    want = torch.stack([RR.apply(rules, logits[b], list(hist)) for b in range(B)])
    got = ops.dec_whisper_rules(logits.clone().cuda(), tokens.cuda(), pos.cuda(), P, eot=EOT, timestamp_begin=TB, no_timestamps=NOTS,
                                max_initial_timestamp=50, suppress=rules.suppress, blank=rules.blank).cpu()
    assert torch.equal(torch.isinf(got), torch.isinf(want)), (torch.isinf(got) != torch.isinf(want)).nonzero()[:5]
    assert torch.equal(got[~torch.isinf(got)], want[~torch.isinf(want)])  # untouched entries are bit-identical
    if torch.isfinite(want).any(-1).all():
        assert torch.equal(got.argmax(-1), want.argmax(-1))


def test_rules_leave_the_forced_prompt_alone():
    from pytorch_models._hip import ops

    logits = torch.randn(2, V)
    tokens = torch.zeros(2, 10, dtype=torch.int64)
    got = ops.dec_whisper_rules(logits.clone().cuda(), tokens.cuda(), torch.tensor([0], dtype=torch.int32).cuda(), 4, eot=EOT,
                                timestamp_begin=TB)
    assert torch.equal(got.cpu(), logits)  # index 1 < P = 4: the prompt is being forced, nothing is filtered


def test_generate_with_rules_produces_well_formed_timestamp_streams():
    """End to end on a small Whisper: greedy ids under the rules (graph replay == eager), first generated token a timestamp
    within the initial cap, timestamps never decreasing, an opening stamp is followed by text or by its partner, suppressed ids
    never appear."""
    from pytorch_models.audio2text import Whisper
    from pytorch_models.audio2text.generate import WhisperRules

    w = Whisper(V, 2, 128)
    fill_module(w, 91)
    bf16_round_(w)
    w = w.to(torch.bfloat16).cuda().eval()
    mel = synth_input("rules_mel", (3, 80, 200), 91).cuda()
    prompt = synth_tokens("rules_prompt", (3, 3), 400, 91).cuda()
    sup = tuple(range(0, 40)) + (NOTS,)
    rules = WhisperRules(eot=EOT, timestamp_begin=TB, no_timestamps=NOTS, max_initial_timestamp=50, suppress=sup, blank=(41, EOT))
    ids = w.generate(mel, prompt, 40, rules=rules).cpu()
    assert torch.equal(ids, w.generate(mel, prompt, 40, rules=rules, graph=False).cpu())
    plain = w.generate(mel, prompt, 40).cpu()
    assert not torch.equal(ids, plain)  # the filters change what an unconstrained random-weight model emits
    for b in range(3):
        gen = ids[b, 3:].tolist()
        assert TB <= gen[0] <= TB + 50
        assert not set(gen) & set(sup) and gen[0] not in (41, EOT)
        stamps = [t for t in gen if t >= TB]
        assert stamps == sorted(stamps)
        for i, t in enumerate(gen[:-1]):
            if t >= TB and (i == 0 or gen[i - 1] >= TB):  # closed pair or lone opening stamp: text follows
                assert gen[i + 1] < TB
            elif t >= TB:  # a stamp after text: the next token closes the segment (a stamp) or ends the transcript
                assert gen[i + 1] >= EOT
