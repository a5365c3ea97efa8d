"""data2vec (audio) on MI355X, drop-in for /root/reference pytorch_models/audio/data2vec_audio.py: the Wav2Vec2
layer-norm stem and post-norm encoder with a five-layer positional conv (grouped Conv1d(k 19) -> LayerNorm1d without
affine -> GELU; data2vec_audio.py:23-30).  Each layer = pm_group_windows + pm_grouped_conv_bf16 (bias in the
epilogue) + one pm_layernorm_ex (no affine, GELU; the last one also adds the "+ x" residual)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .._hip import ops
from ..transformer import Encoder, LayerNorm
from .wav2vec2 import LayerNorm1d, Wav2Vec2


class Data2VecAudio(Wav2Vec2):
    PE_KERNEL = 19
    _HF_FLAVOUR = "data2vec"  # five plain positional convs instead of one weight-normed (data2vec_audio.py:55-56)

    def __init__(self, n_layers: int, d_model: int, stem_bias: bool = False, dropout: float = 0.0) -> None:
        nn.Module.__init__(self)
        self._front(d_model, stem_bias, False, dropout)  # layer-norm stem
        conv = lambda: nn.Conv1d(d_model, d_model, self.PE_KERNEL, padding=self.PE_KERNEL // 2, groups=self.PE_GROUPS)
        self.pe_conv = nn.Sequential(*[nn.Sequential(conv(), LayerNorm1d(d_model, elementwise_affine=False), nn.GELU()) for _ in range(5)])
        self.layers = Encoder(n_layers, d_model, dropout=dropout, pre_norm=False)
        self.norm = LayerNorm(d_model)
        self.pre_norm = False

    def forward(self, x: Tensor) -> Tensor:
        h0 = self._features(x)
        B, T, d = h0.shape
        h = h0
        for i, blk in enumerate(self.pe_conv):
            conv, norm = blk[0], blk[1]
            p = conv.padding[0]
            y = self.grouped_conv(conv, h, (p, p), "none", None)
            last = i == len(self.pe_conv) - 1
            h = ops.layernorm(y.view(B * T, d), None, None, norm.eps, act="gelu", resid=h0.view(B * T, d) if last else None).view(B, T, d)
        return self.layers(self.norm(h)).to(self.norm.weight.dtype)
