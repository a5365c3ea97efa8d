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


class OutputGatherer:
    """The gather of one fixed output geometry with every buffer allocated ONCE: the padded send buffer, the receive
    buffer and (ragged shards only) the trimmed result.  ``__call__`` is one copy into the send buffer (skipped when the
    caller writes its outputs straight into ``send_view()``) and one ``all_gather_into_tensor``; with even shards the
    receive buffer IS the ordered result, so nothing else runs per call."""

    def __init__(self, n_total: int, tail: tuple, dtype: torch.dtype, device, group=None) -> None:
        self.group, self.n_total = group, n_total
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.per = -(-n_total // self.world)
        self.lo, self.hi = shard_bounds(n_total, self.rank, self.world)
        self.even = n_total % self.world == 0
        self.send = torch.zeros((self.per,) + tuple(tail), dtype=dtype, device=device)
        self.recv = torch.empty((self.world * self.per,) + tuple(tail), dtype=dtype, device=device)
        self.out = self.recv if self.even else torch.empty((n_total,) + tuple(tail), dtype=dtype, device=device)

    def send_view(self) -> Tensor:
        """This rank's rows of the send buffer: a model may write its outputs here directly."""
        return self.send[: self.hi - self.lo]

    def __call__(self, local: Tensor | None = None) -> Tensor:
        if local is not None and local.data_ptr() != self.send.data_ptr():
            self.send[: local.shape[0]].copy_(local)
        if self.world == 1:
            return self.send[: self.n_total]
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        if self.even:
            return self.recv
        for r in range(self.world):
            lo, hi = shard_bounds(self.n_total, r, self.world)
            self.out[lo:hi].copy_(self.recv[r * self.per : r * self.per + (hi - lo)])
        return self.out


def gather_outputs(local: Tensor, n_total: int, group=None, ws: OutputGatherer | None = None) -> Tensor:
    """all_gather per-rank outputs (ragged shards are padded to the largest shard and trimmed) -> (n_total, ...).
    ``ws``: a preallocated :class:`OutputGatherer` of this geometry (steady-state loops: no allocation per call)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if ws is None:
        ws = OutputGatherer(n_total, tuple(local.shape[1:]), local.dtype, local.device, group)
    return ws(local)


def run_dp(fn: Callable[[Tensor], Tensor], batch: Tensor, group=None, ws: OutputGatherer | None = None) -> Tensor:
    """Every rank holds the full `batch` (or at least its own shard's rows); rank r computes fn on its contiguous
    shard and all ranks return the full, ordered output."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    return gather_outputs(fn(batch[lo:hi]), batch.shape[0], group, ws)
