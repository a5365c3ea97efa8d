"""BERT / RoBERTa encoder on MI355X: drop-in for /root/reference pytorch_models/text/bert.py (BERT, from_hf,
load_hf_state_dict; parameter names token_embs, pos_embs, norm, layers).  Token-type embeddings, pooler and classifier
are not part of the reference class either (bert.py:14); the token-type row 0 is merged into pos_embs at load time."""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from .._hip import ops
from ..transformer import Encoder, LayerNorm, _f32, _wb


class BERT(nn.Module):
    def __init__(self, vocab_size: int, n_layers: int, d_model: int, max_seq_len: int = 512, dropout: float = 0.0,
                 norm_eps: float = 1e-12) -> None:
        super().__init__()
        vocab_size = math.ceil(vocab_size / 64) * 64  # padded to a multiple of 64, as the reference does (bert.py:29)
        self.token_embs = nn.Embedding(vocab_size, d_model)
        self.pos_embs = nn.Parameter(torch.zeros(max_seq_len, d_model))
        self.norm = LayerNorm(d_model, norm_eps)
        self.layers = Encoder(n_layers, d_model, dropout=dropout, pre_norm=False, norm_eps=norm_eps)

    def forward(self, x: Tensor) -> Tensor:
        """token ids (..., L) int64 -> hidden states (..., L, d) bf16 (bert.py:35-40)."""
        E = self.token_embs.weight  # fp32 table -> fp32 rows (and fp32 blocks behind them), bf16 -> bf16
        lead = x.shape
        h = ops.embed_tokens(x.reshape(-1, lead[-1]), E, _f32(self, "pos", self.pos_embs))
        h = self.layers(self.norm(h))
        return h.view(*lead, h.shape[-1])

    @staticmethod
    def from_config(config: dict, **kwargs) -> "BERT":
        """A Hugging Face config.json as a dict (the reference downloads it: bert.py:44-58; here it is passed in)."""
        max_pos = config["max_position_embeddings"]
        if "roberta" in config["model_type"]:  # RoBERTa never uses its first two position rows (bert.py:56-58)
            max_pos -= 2
        return BERT(vocab_size=config["vocab_size"], n_layers=config["num_hidden_layers"], d_model=config["hidden_size"],
                    max_seq_len=max_pos, norm_eps=config["layer_norm_eps"], **kwargs)

    @staticmethod
    def from_hf(model_tag: str, *, pretrained: bool = False, **kwargs) -> "BERT":
        raise NotImplementedError(
            f"BERT.from_hf({model_tag!r}) fetches config.json (and weights) from the network, which this build does not do; "
            "use BERT.from_config(json.load(open('config.json'))) and load_hf_state_dict(torch.load(path, weights_only=True)).")

    def load_hf_state_dict(self, state_dict: dict[str, Tensor]) -> None:
        """Hugging Face BertModel / RobertaModel state_dict (converters.load_hf_bert; placement as bert.py:79-107)."""
        from ..converters import load_hf_bert

        load_hf_bert(self, state_dict)
