"""GPT (the 2018 model) on MI355X: drop-in for /root/reference pytorch_models/text/gpt.py (GPT, from_openai; parameter
names token_embs, pos_embs, layers).  Post-norm causal decoder with tanh-GELU, no final norm, tied logits."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from ..transformer import Decoder
from .gpt2 import _lm_forward


class GPT(nn.Module):
    vocab_size = 40478
    max_seq_len: int = 512

    def __init__(self, n_layers: int = 12, d_model: int = 768, dropout: float = 0.0) -> None:
        super().__init__()
        self.token_embs = nn.Embedding(self.vocab_size, d_model)
        self.pos_embs = nn.Parameter(torch.zeros(self.max_seq_len, d_model))
        self.layers = Decoder(n_layers, d_model, dropout=dropout, pre_norm=False, act="approximate_gelu")

    def forward(self, x: Tensor) -> Tensor:
        """token ids (..., L) int64 -> logits (..., L, 40478) fp32 (gpt.py:24-29)."""
        return _lm_forward(self, x, None)

    @staticmethod
    def from_openai(*, pretrained=False, **kwargs) -> "GPT":
        m = GPT(**kwargs)
        if pretrained:
            raise NotImplementedError(
                "GPT.from_openai(pretrained=True) needs a network download, which this build does not do; construct with "
                "pretrained=False and call load_openai_params(list of arrays in params_shapes.json order).")
        return m

    def load_openai_params(self, params) -> None:
        """The flat parameter list of openai/finetune-transformer-lm, already split and reshaped as gpt.py:40-52 does
        (converters.load_openai_gpt; placement as gpt.py:54-84)."""
        from ..converters import load_openai_gpt

        load_openai_gpt(self, params)
