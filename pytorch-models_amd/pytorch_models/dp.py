"""Data-parallel inference over the GPUs of one node: one process per GPU, contiguous batch shards,
replicated weights, ONE all_gather of the outputs (RCCL over xGMI when the backend is "nccl").

The reference has no distributed code at all; samples are independent on this path (even the
log-mel peak is per sample, whisper.py:146), so there is no data-path collective - only the final gather.
Payloads are tiny (ViT: (N, d) bf16; Whisper: (N, T) int64 ids), i.e. latency-bound, not link-bound.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist
from torch import Tensor


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n samples for `rank`: the first n % world ranks take one extra sample."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(batch: Tensor, rank: int | None = None, world: int | None = None) -> Tensor:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    return batch[lo:hi]


def gather_outputs(local: Tensor, n_total: int, group=None) -> Tensor:
    """all_gather per-rank outputs (ragged shards are padded to the largest shard and trimmed) -> (n_total, ...)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    per = -(-n_total // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        parts.append(out[r * per : r * per + (hi - lo)])
    return torch.cat(parts, 0)


def run_dp(fn: Callable[[Tensor], Tensor], batch: Tensor, group=None) -> Tensor:
    """Every rank holds the full `batch` (or at least its own shard's rows); rank r computes fn on its contiguous
    shard and all ranks return the full, ordered output."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    return gather_outputs(fn(batch[lo:hi]), batch.shape[0], group)
