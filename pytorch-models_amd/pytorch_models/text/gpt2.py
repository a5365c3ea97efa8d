"""GPT-2 on MI355X: drop-in for /root/reference pytorch_models/text/gpt2.py (GPT2, from_hf, load_hf_state_dict; parameter
names token_embs, pos_embs, layers, norm).

forward = pm_embed_tokens (token + position rows in one kernel) -> Decoder of pre-norm, causal, tanh-GELU layers
(transformer.py) -> LayerNorm -> logits against the tied embedding matrix (pm_linear_bf16, fp32 out, ragged N = 50257).
Greedy generation with a KV cache: GPT2.generate / text.DecoderGenerator (generate.py's decode-step kernels)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .._hip import ops
from ..transformer import Decoder, LayerNorm, _f32, _wb

_SIZES = {"gpt2": (12, 768), "gpt2-medium": (24, 1024), "gpt2-large": (36, 1280), "gpt2-xl": (48, 1600)}


def _lm_forward(m, tokens: Tensor, final_norm) -> Tensor:
    lead = tokens.shape
    tok2 = tokens.reshape(-1, lead[-1])  # the reference also accepts an unbatched (L,) sequence (generator.py:25)
    if m.token_embs.weight.dtype == torch.float32:  # fp32 parameters: fp32 rows, fp32 blocks, fp32 logits
        E = m.token_embs.weight
        h = m.layers(ops.embed_tokens(tok2, E, m.pos_embs))
        if final_norm is not None:
            h = final_norm(h)
        return ops.linear_f32(h.view(-1, h.shape[-1]), E).view(*lead, E.shape[0])
    E = m.token_embs.weight
    h = ops.embed_tokens(tok2, E, _f32(m, "pos", m.pos_embs))
    h = m.layers(h)
    if final_norm is not None:
        h = final_norm(h)
    logits = ops.linear(h.view(-1, h.shape[-1]), E, None, out_dtype=torch.float32)
    return logits.view(*lead, E.shape[0])


class GPT2(nn.Module):
    vocab_size = 50257
    max_seq_len: int = 1024

    def __init__(self, n_layers: int, d_model: int, dropout: float = 0.0) -> None:
        super().__init__()
        self.token_embs = nn.Embedding(self.vocab_size, d_model)
        self.pos_embs = nn.Parameter(torch.zeros(self.max_seq_len, d_model))
        self.layers = Decoder(n_layers, d_model, dropout=dropout, act="approximate_gelu")
        self.norm = LayerNorm(d_model)

    def forward(self, x: Tensor) -> Tensor:
        """token ids (..., L) int64 -> logits (..., L, 50257) fp32 (gpt2.py:21-27)."""
        return _lm_forward(self, x, self.norm)

    @torch.no_grad()
    def generate(self, prompt: Tensor, max_new_tokens: int, *, graph: bool = True, topk: int = 1, seed: int = 0,
                 path: str = "auto") -> Tensor:
        """Batched decoding with a KV cache: (B, P) int64 prompt -> (B, P + max_new_tokens) ids; greedy (topk = 1) or
        top-k sampling on the device (softmax over the k largest logits; the same seed gives the same ids)."""
        from ..audio2text.generate import greedy_decode, greedy_exact

        if self.token_embs.weight.dtype == torch.float32 and topk == 1:  # fp32 parameters: fp32 end to end
            return greedy_exact(self, None, prompt, max_new_tokens)
        return greedy_decode(self, None, prompt, max_new_tokens, graph=graph, topk=topk, seed=seed, path=path)

    @staticmethod
    def from_hf(model_tag: str, *, pretrained=False, **kwargs) -> "GPT2":
        n_layers, d_model = _SIZES[model_tag]
        m = GPT2(n_layers, d_model, **kwargs)
        if pretrained:
            raise NotImplementedError(
                "GPT2.from_hf(pretrained=True) needs a network download, which this build does not do; construct with "
                "pretrained=False and call load_hf_state_dict(torch.load(path, weights_only=True)).")
        return m

    def load_hf_state_dict(self, state_dict: dict[str, Tensor]) -> None:
        """Hugging Face GPT2LMHeadModel state_dict (converters.load_hf_gpt2)."""
        from ..converters import load_hf_gpt2

        load_hf_gpt2(self, state_dict)
