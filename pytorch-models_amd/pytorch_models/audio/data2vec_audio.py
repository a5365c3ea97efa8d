"""data2vec (audio) on MI355X, drop-in for /root/reference pytorch_models/audio/data2vec_audio.py: the Wav2Vec2
layer-norm stem and post-norm encoder with a five-layer positional conv (grouped Conv1d(k 19) -> LayerNorm1d without
affine -> GELU; data2vec_audio.py:23-30).  Each layer = pm_group_windows + pm_grouped_conv_bf16 (bias in the
epilogue) + one pm_layernorm_ex (no affine, GELU; the last one also adds the "+ x" residual)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .._hip import ops
from ..transformer import Encoder, LayerNorm, Linear
from .wav2vec2 import FeatureEncoder, LayerNorm1d, Wav2Vec2


class Data2VecAudio(Wav2Vec2):
    PE_KERNEL = 19

    def __init__(self, n_layers: int, d_model: int, stem_bias: bool = False, dropout: float = 0.0) -> None:
        nn.Module.__init__(self)
        self.feature_encoder = FeatureEncoder(self.STEM_DIMS, self.STEM_KERNELS, self.STEM_STRIDES, stem_bias, dropout)
        in_dim = self.STEM_DIMS[-1]
        self.proj = nn.Sequential(LayerNorm(in_dim))
        if in_dim != d_model:
            self.proj.append(Linear(in_dim, d_model))
        self.pe_conv = nn.Sequential()
        for _ in range(5):
            self.pe_conv.append(nn.Sequential(
                nn.Conv1d(d_model, d_model, self.PE_KERNEL, padding=self.PE_KERNEL // 2, groups=self.PE_GROUPS),
                LayerNorm1d(d_model, elementwise_affine=False),
                nn.GELU(),
            ))
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

    @torch.no_grad()
    def load_hf_state_dict(self, state_dict: dict[str, Tensor]) -> None:
        sd = dict(state_dict)
        self._load_stem_and_layers(sd, "feature_projection.layer_norm", "feature_projection.projection")
        for i, blk in enumerate(self.pe_conv):
            blk[0].weight.copy_(sd.pop(f"encoder.pos_conv_embed.layers.{i}.conv.weight"))
            blk[0].bias.copy_(sd.pop(f"encoder.pos_conv_embed.layers.{i}.conv.bias"))
        print(sd.keys())
