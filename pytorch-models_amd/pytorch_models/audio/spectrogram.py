"""Audio front end on MI355X, drop-in for /root/reference pytorch_models/audio/spectrogram.py
(Spectrogram, get_mel_filters, MelSpectrogram; ``window`` non-persistent and ``filters`` persistent
buffers as in the reference, spectrogram.py:12,41).

The STFT is a framed real DFT on the f32 MFMA (exact fp32 FMA chains) with reflect padding and the
Hann window folded into the twiddle table; see csrc/logmel.hip.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from .. import _cpu
from .._hip import ops
from ..transformer import derived


def get_mel_filters(n_mels: int, n_fft: int, sample_rate: float) -> Tensor:
    """Slaney-scale, Slaney-normalised triangular filterbank (n_mels, n_fft // 2 + 1), float32.

    Mel scale: linear (200/3 Hz per mel) below 1 kHz = 15 mel, logarithmic above with 27 mel per
    factor 6.4.  Computed in float64 on the host and rounded once."""
    nyq = sample_rate / 2
    top = nyq * 3 / 200 if nyq < 1000 else 15 + 27 * math.log(nyq / 1000, 6.4)
    mel = torch.linspace(0, top, n_mels + 2, dtype=torch.float64)
    edge = torch.where(mel < 15, mel * (200 / 3), 1000 * 6.4 ** ((mel - 15) / 27))  # Hz of each band edge
    bins = torch.linspace(0, nyq, n_fft // 2 + 1, dtype=torch.float64)
    up = (bins[None, :] - edge[:-2, None]) / (edge[1:-1] - edge[:-2])[:, None]
    down = (edge[2:, None] - bins[None, :]) / (edge[2:] - edge[1:-1])[:, None]
    tri = torch.minimum(up, down).clamp_(min=0)
    tri *= (2 / (edge[2:] - edge[:-2]))[:, None]
    return tri.float()


class Spectrogram(nn.Module):
    def __init__(self, n_fft: int, hop_length: int) -> None:
        super().__init__()
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.register_buffer("window", torch.hann_window(n_fft), False)
        self.window: Tensor

    def _tables(self):
        return derived(self, "tw", (self.window,), lambda: ops.stft_tables(self.window, self.n_fft))

    def forward(self, x: Tensor) -> Tensor:
        """(..., T) -> (..., n_fft/2+1, 1 + T // hop) power spectrogram (center=True, reflect padding)."""
        if _cpu.on_cpu(x, self.window):
            return _cpu.spectrogram(x, self.window, self.n_fft, self.hop_length)
        return ops.stft_mel(x, self._tables(), self.n_fft, self.hop_length, 1 + x.shape[-1] // self.hop_length, 0)


class MelSpectrogram(Spectrogram):
    def __init__(self, n_fft: int, hop_length: int, n_mels: int, sample_rate: int) -> None:
        super().__init__(n_fft, hop_length)
        self.register_buffer("filters", get_mel_filters(n_mels, n_fft, sample_rate))
        self.filters: Tensor

    def _csr(self):
        return derived(self, "csr", (self.filters,), lambda: ops.mel_csr(self.filters))

    def forward(self, x: Tensor) -> Tensor:
        if _cpu.on_cpu(x, self.window, self.filters):
            return _cpu.mel_spectrogram(x, self.window, self.filters, self.n_fft, self.hop_length)
        return ops.stft_mel(x, self._tables(), self.n_fft, self.hop_length, 1 + x.shape[-1] // self.hop_length, 1,
                            self._csr(), self.filters.shape[0])
