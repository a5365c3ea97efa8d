"""Oracle for Whisper's decoding-time logit filters (test infrastructure, see oracle/__init__.py).

PARITY UNPINNED: /root/reference has no Whisper decoding (README.md:86-87 lists the tokenizer and timestamp handling as TODO)
and OpenAI's ``whisper`` package is not in this image, so there is neither a reference implementation nor a fixture to pin this
file to.  It restates, in plain Python over one sequence at a time, the published behaviour of openai/whisper decoding.py
(SuppressTokens, SuppressBlank, ApplyTimestampRules): the HIP kernel pm_dec_whisper_rules is tested against THIS restatement
and against invariants of the produced token streams (timestamps in non-decreasing pairs, first token a timestamp).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch
from torch import Tensor

NEG = float("-inf")


@dataclass
class Rules:
    eot: int
    timestamp_begin: int
    no_timestamps: int = -1
    max_initial_timestamp: int = -1  # index relative to timestamp_begin; < 0 = no cap
    suppress: list = field(default_factory=list)
    blank: list = field(default_factory=list)  # ids forbidden as the FIRST generated token (blank, eot)


def apply(rules: Rules, logits: Tensor, generated: list[int]) -> Tensor:
    """One sequence: logits (V,) f32 of the position being chosen, ``generated`` = ids produced so far (prompt excluded)."""
    lg = logits.clone().float()
    tb = rules.timestamp_begin
    for i in rules.suppress:
        lg[i] = NEG
    if not generated:
        for i in rules.blank:
            lg[i] = NEG
    if rules.no_timestamps >= 0:
        lg[rules.no_timestamps] = NEG
    last_was_ts = len(generated) >= 1 and generated[-1] >= tb
    penultimate_was_ts = len(generated) < 2 or generated[-2] >= tb
    if last_was_ts:
        if penultimate_was_ts:  # a closed pair (or a lone opening stamp): text must follow
            lg[tb:] = NEG
        else:  # an open segment: it must be closed (or the transcript ended) before more text
            lg[: rules.eot] = NEG
    stamps = [t for t in generated if t >= tb]
    if stamps:  # never go back in time; a new segment must have a non-zero length
        floor = stamps[-1] if (last_was_ts and not penultimate_was_ts) else stamps[-1] + 1
        lg[tb:floor] = NEG
    if not generated:
        lg[:tb] = NEG
        if rules.max_initial_timestamp >= 0:
            lg[tb + rules.max_initial_timestamp + 1:] = NEG
    logprobs = torch.log_softmax(lg, -1)
    if torch.isfinite(logprobs[tb:]).any():
        if torch.logsumexp(logprobs[tb:], -1) > logprobs[:tb].max():
            lg[:tb] = NEG
    return lg
