"""SEW (squeezed and efficient wav2vec) on MI355X, drop-in for /root/reference pytorch_models/audio/sew.py: a slimmer
13-layer legacy stem, the transformer running at half the frame rate (avg-pool + stride-2 positional conv on the way
in, a Linear(d, 2d) + GELU whose output IS the up-sampled sequence on the way out)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .._hip import ops
from ..transformer import Linear, _f32, _wb
from .wav2vec2 import Wav2Vec2


class SEW(Wav2Vec2):
    STEM_DIMS = (64,) + (128,) * 4 + (256,) * 4 + (512,) * 4
    STEM_KERNELS = (10,) + (3, 1) * 4 + (2, 1) * 2
    STEM_STRIDES = (5,) + (2, 1) * 6

    PE_KERNEL = 31
    _HF_FLAVOUR = "sew"  # projection keys without the feature_projection prefix, plus encoder.upsample (sew.py:55-79)

    def __init__(self, n_layers: int, d_model: int, stem_bias: bool = True, stem_legacy: bool = True, dropout: float = 0.0) -> None:
        assert stem_legacy
        super().__init__(n_layers, d_model, stem_bias, stem_legacy, dropout, False)
        self.pe_conv[1].stride = (2,)
        self.upsample = nn.Sequential(Linear(d_model, d_model * 2), nn.GELU())

    def forward(self, x: Tensor) -> Tensor:
        """waveform (B, L) -> (B, T, d) at the stem's frame rate."""
        h = self._features(x)
        B, T, d = h.shape
        pad = self.pe_conv[0].padding
        h = self.grouped_conv(self.pe_conv[1], h, (pad[0], pad[1]), "gelu", ops.avgpool_time2(h))  # avg_pool(x) + GELU(conv(x))
        h = self.layers(self.norm(h))
        up = self.upsample[0]
        y = ops.linear(h.reshape(-1, d), _wb(up, "w", up.weight), _f32(up, "b", up.bias), act="gelu")  # (B*T/2, 2d) row-major == (B, T/2 * 2, d)
        y = y.view(B, -1, d)
        if y.shape[1] < T:  # odd T: the dropped last frame comes back as zeros (sew.py:37-38)
            y = torch.cat([y, y.new_zeros(B, T - y.shape[1], d)], 1)
        return y.to(self.norm.weight.dtype)
