"""CPU, world_size 2, gloo: the N > 1 path of the data-parallel wrapper (shard -> compute -> all_gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytorch_models import dp


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 5, 8, 256, 2047):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
        seen = []

        def fn(x):  # stands in for model(x): per-sample, order-preserving
            seen.append(x.shape[0])
            return torch.stack([x.sum(1), x[:, 0] * 2], 1)

        out = dp.run_dp(fn, batch)
        ids = dp.run_dp(lambda x: (x[:, :1] * 10).long().expand(-1, 4).contiguous(), batch)  # int64 "token ids"
        q.put((rank, out, ids, seen[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 5, 1])
def test_run_dp_world2_gloo(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    batch = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
    want = torch.stack([batch.sum(1), batch[:, 0] * 2], 1)
    sizes = {}
    for rank, out, ids, seen in got:
        torch.testing.assert_close(out, want, rtol=0, atol=0)  # every rank holds the full ordered output
        assert torch.equal(ids, (batch[:, :1] * 10).long().expand(-1, 4))
        sizes[rank] = seen
    assert sizes[0] + sizes[1] == n and sizes[0] - sizes[1] in (0, 1)
