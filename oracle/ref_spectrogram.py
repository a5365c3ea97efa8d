"""Oracle for the audio front end (test infrastructure, see oracle/__init__.py).

Restates /root/reference pytorch_models/audio/spectrogram.py:7-45 and
pytorch_models/audio2text/whisper.py:138-148 without ``torch.stft``: explicit reflect padding,
framing, Hann window and a real DFT (fp64 matmul by default, ``rfft`` for the timed baseline).
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import Tensor


def hann_window(n: int) -> Tensor:
    """torch.hann_window(n) (periodic) - spectrogram.py:12: 0.5 - 0.5 cos(2 pi i / n)."""
    i = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * i / n)).float()


def frames(x: Tensor, n_fft: int, hop: int) -> Tensor:
    """center=True, pad_mode="reflect" framing of torch.stft - spectrogram.py:16.
    (..., T) -> (..., 1 + T // hop, n_fft); frame t covers padded samples [t*hop, t*hop + n_fft)."""
    pad = n_fft // 2
    lead = x.shape[:-1]
    xp = torch.nn.functional.pad(x.reshape(-1, 1, x.shape[-1]), (pad, pad), mode="reflect").squeeze(1)
    return xp.unfold(-1, n_fft, hop).reshape(*lead, -1, n_fft)


def power_spectrogram(x: Tensor, n_fft: int, hop: int, dft: str = "matmul") -> Tensor:
    """Spectrogram.forward - spectrogram.py:15-16: |STFT|^2, onesided, (..., n_fft/2+1, n_frames)."""
    fr = frames(x, n_fft, hop) * hann_window(n_fft)
    if dft == "rfft":
        X = torch.fft.rfft(fr, dim=-1)
        p = X.real.square() + X.imag.square()
    else:
        n = torch.arange(n_fft, dtype=torch.float64)
        k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)
        ang = 2.0 * math.pi * torch.outer(n, k) / n_fft
        f64 = fr.double()
        p = ((f64 @ torch.cos(ang)).square() + (f64 @ torch.sin(ang)).square()).float()
    return p.transpose(-1, -2)


def mel_filters(n_mels: int, n_fft: int, sample_rate: float) -> Tensor:
    """get_mel_filters - spectrogram.py:19-35: Slaney mel scale (linear below 1 kHz = 15 mel, log with
    base 6.4 per 27 mel above), triangular filters, Slaney area normalisation 2 / (f_hi - f_lo)."""
    f_max = sample_rate / 2
    mel_max = f_max * 3 / 200 if f_max < 1000 else 15 + 27 * math.log(f_max / 1000, 6.4)
    mel = np.linspace(0.0, mel_max, n_mels + 2)
    hz = np.where(mel < 15, mel * 200 / 3, 1000 * 6.4 ** ((mel - 15) / 27))
    fft_hz = np.linspace(0.0, sample_rate / 2, n_fft // 2 + 1)
    width = np.diff(hz)
    ramp = hz[:, None] - fft_hz[None, :]
    lower = -ramp[:-2] / width[:-1, None]
    upper = ramp[2:] / width[1:, None]
    tri = np.clip(np.minimum(lower, upper), 0.0, None)
    tri *= (2.0 / (hz[2:] - hz[:-2]))[:, None]
    return torch.from_numpy(tri.astype(np.float32))


def mel_spectrogram(x: Tensor, n_fft: int, hop: int, n_mels: int, sample_rate: int, dft: str = "matmul") -> Tensor:
    """MelSpectrogram.forward - spectrogram.py:44-45: filters @ power."""
    return mel_filters(n_mels, n_fft, sample_rate) @ power_spectrogram(x, n_fft, hop, dft)


def whisper_log_mel(x: Tensor, n_mels: int = 80, dft: str = "matmul") -> Tensor:
    """WhisperPreprocessor.forward - whisper.py:143-148: drop the last frame, log10 of the clamped mel
    power, floor at (PER-SAMPLE max - 8) (SURVEY.md finding F4), then (x + 4) / 4."""
    m = mel_spectrogram(x, 400, 160, n_mels, 16_000, dft)[..., :-1]
    m = m.clamp(min=0).log10()
    peak = m.flatten(-2).max(-1).values[..., None, None]
    m = torch.maximum(m, peak - 8)
    return (m + 4) / 4
