"""Deterministic, torch-RNG-independent synthetic weights and inputs.

Every tensor of a ``state_dict`` is filled from ``numpy.random.PCG64`` seeded by
``crc32(key) ^ seed``, so any machine regenerates bit-identical weights for an
86 M .. 1.5 B parameter model without shipping them.  The same generator feeds

* ``tests/golden/make_golden.py`` (run once against the reference import),
* the parity tests (oracle and HIP path get the same tensors), and
* ``bench.py`` (synthetic weights of the benchmarked geometry).

This module is deliberately free of any ``pytorch_models`` import so it can be
used next to either the reference package or this repo's package.

Unlike the reference's default init (zeros for ``pe`` / ``cls_token`` /
``pos_embs``, ones/zeros for LayerNorm: /root/reference pytorch_models/image/vit.py:65-66,
audio2text/whisper.py:24,39) every tensor gets non-trivial values so that a
dropped term cannot hide behind a zero.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFFFFFF))


def _std_for(key: str, shape: tuple[int, ...]) -> tuple[float, float]:
    """(mean, std) of the normal used for ``key``."""
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "weight" and len(shape) >= 2:  # Linear / Conv / Embedding
        fan_in = int(np.prod(shape[1:]))
        return 0.0, 1.0 / np.sqrt(fan_in)
    if leaf == "weight":  # LayerNorm gain
        return 1.0, 0.1
    if leaf == "bias":
        return 0.0, 0.05
    if leaf == "running_var":  # BatchNorm: a variance is positive
        return 1.0, 0.1
    # pe, cls_token, probe, pos_embs and other free parameters / buffers
    return 0.0, 0.1


def synth_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    mean, std = _std_for(key, shape)
    arr = _rng(key, seed).standard_normal(shape, dtype=np.float32)
    arr *= np.float32(std)
    if mean != 0.0:
        arr += np.float32(mean)
    return torch.from_numpy(arr)


@torch.no_grad()
def fill_module(module: torch.nn.Module, seed: int = 0, skip: tuple[str, ...] = ("window", "filters")) -> None:
    """Overwrite every floating parameter and buffer of ``module`` in place (fp32 values,
    cast to the tensor's dtype/device).  Keys whose leaf name is in ``skip`` keep their value
    (the STFT window and mel filterbank are analytic, not learned)."""
    named = dict(module.named_parameters())
    named.update({k: v for k, v in module.named_buffers()})
    for key in sorted(named):
        t = named[key]
        if not t.is_floating_point() or key.rsplit(".", 1)[-1] in skip:
            continue
        t.copy_(synth_tensor(key, t.shape, seed).to(device=t.device, dtype=t.dtype))


def synth_input(tag: str, shape, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    """N(0, scale^2) fp32 input tensor keyed by ``tag`` (images, waveforms, mel frames)."""
    arr = _rng("input:" + tag, seed).standard_normal(tuple(int(s) for s in shape), dtype=np.float32)
    arr *= np.float32(scale)
    return torch.from_numpy(arr)


def synth_tokens(tag: str, shape, vocab: int, seed: int = 0) -> torch.Tensor:
    arr = _rng("tokens:" + tag, seed).integers(0, vocab, size=tuple(int(s) for s in shape), dtype=np.int64)
    return torch.from_numpy(arr)


def bf16_round_(module_or_sd) -> None:
    """Round every floating tensor to the nearest bf16 value, keeping fp32 storage.
    Gives the oracle exactly the weights the bf16 HIP path sees."""
    items = module_or_sd.items() if isinstance(module_or_sd, dict) else module_or_sd.state_dict().items()
    with torch.no_grad():
        for _, t in items:
            if t.is_floating_point():
                t.copy_(t.to(torch.bfloat16).to(t.dtype))
