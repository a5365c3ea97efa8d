"""Text generation for decoder-only models: drop-in for /root/reference pytorch_models/text/generator.py
(DecoderGenerator(model, tokenizer).generate(prompt, max_tokens, topk)).

A pre-norm model (GPT-2) runs the KV-cached decode-step kernels of audio2text/generate.py, greedy or with top-k
sampling on the device (pm_dec_sample_topk) - the reference re-runs the whole sequence for every new token
(generator.py:24-25, O(T^2)); post-norm models (GPT) re-run the HIP forward per token like the reference does.  The tokenizer is any object with
``encode(str) -> list[int]``, ``decode(list[int]) -> str`` and ``eos_token_id`` (no tokenizer ships with this build)."""
from __future__ import annotations

import torch
from torch import nn


class DecoderGenerator:
    def __init__(self, model: nn.Module, tokenizer) -> None:
        self.model = model
        self.tokenizer = tokenizer

    @torch.inference_mode()
    def generate_ids(self, tokens: list[int], max_tokens: int = 100, topk: int = 1, eos_token_id: int | None = None,
                     generator: torch.Generator | None = None, seed: int = 0) -> list[int]:
        """Token-level form of generate(): prompt ids -> prompt + new ids, stopping after eos_token_id (kept, as the
        reference keeps it) or max_tokens new tokens."""
        p0 = next(self.model.parameters())
        device = p0.device
        tokens = list(tokens)
        n = len(tokens)
        # KV-cached decoding: the graph-replayed step kernels for bf16 pre-norm stacks (GPT-2), the fp32 cached loop
        # (generate.greedy_exact) for everything else the layer algebra covers - fp32 parameters, post-norm stacks (GPT) -
        # instead of the reference's O(T^2) full-prefix loop, which remains for top-k sampling on those models
        kv_ok = p0.dtype == torch.bfloat16 and all(l.pre_norm for l in self.model.layers)
        # a model without a position table (anything that is not one of this package's decoders) has no length limit here
        room = self.model.pos_embs.shape[0] - n if hasattr(self.model, "pos_embs") else max_tokens
        if topk <= 64 and hasattr(self.model, "generate") and kv_ok:
            out = self.model.generate(torch.tensor([tokens], device=device), min(max_tokens, room), topk=topk, seed=seed)[0].tolist()
            new = out[n:]
            if eos_token_id is not None and eos_token_id in new:
                new = new[: new.index(eos_token_id) + 1]
            return tokens + new
        if topk == 1 and hasattr(self.model, "token_embs") and hasattr(self.model, "pos_embs"):
            from ..audio2text.generate import greedy_exact
            from ..transformer import derived

            m32 = self.model
            if p0.dtype != torch.float32:  # fp32 twin of a bf16 post-norm model (same values), rebuilt when a parameter changes
                import copy

                def build():
                    twin = copy.deepcopy(self.model).float()
                    for mod in twin.modules():
                        mod.__dict__.pop("_pm_derived", None)
                    return twin

                m32 = derived(self.model, "exact32", list(self.model.parameters()), build)
            out = greedy_exact(m32, None, torch.tensor([tokens], device=device), min(max_tokens, room))[0].tolist()
            new = out[n:]
            if eos_token_id is not None and eos_token_id in new:
                new = new[: new.index(eos_token_id) + 1]
            return tokens + new
        while len(tokens) - n < max_tokens:
            logits = self.model(torch.tensor(tokens, device=device))[-1]
            if topk == 1:
                token = int(logits.argmax(-1))
            else:  # top-k sampling (generator.py:30-32)
                vals, idx = logits.topk(topk)
                token = int(idx[torch.multinomial(vals.softmax(-1), 1, generator=generator)])
            tokens.append(token)
            if eos_token_id is not None and token == eos_token_id:
                break
        return tokens

    def generate(self, prompt: str, max_tokens: int = 100, topk: int = 1) -> str:
        ids = self.generate_ids(self.tokenizer.encode(prompt), max_tokens, topk, getattr(self.tokenizer, "eos_token_id", None))
        return self.tokenizer.decode(ids)
