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


def _run_bench(extra, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "stub", "--backend", "gloo",
                        "--no-cpu-baseline", "--steps", "3", "--warmup", "1"] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


def test_bench_py_starts_its_own_ranks_and_times_the_product_gather():
    """`python bench.py --gpus 2` with no torch.distributed environment (how a scaling driver may call it): bench.py starts
    the two ranks itself, every step ends in dp.OutputGatherer (the product's gather, preallocated buffers), rank 0 prints
    one line with the whole-job aggregate.  The stub step stands in for the model: same file, same N > 1 code path."""
    res = _run_bench(["--gpus", "2", "--batch", "5"])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 10 and res["config"]["parallelism"] == "dp2"
    assert res["gather_ok"] is True  # every rank's rows, in rank order
    assert res["value"] > 0 and abs(res["value"] - 10 * 3 / (res["ms_per_step"] * 3e-3)) / res["value"] < 1e-2


def test_bench_py_single_rank_line():
    res = _run_bench(["--gpus", "1"])
    assert res["n_gpus"] == 1 and res["config"]["collective"] == "none" and res["gather_ok"] is True


def _gatherer_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = dp.OutputGatherer(n, (3,), torch.float32, "cpu")
        lo, hi = dp.shard_bounds(n, rank, world)
        full = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
        ptrs = (g.send.data_ptr(), g.recv.data_ptr(), g.out.data_ptr())
        outs = []
        for it in range(3):  # steady state: same buffers every call, results track the inputs
            outs.append(g(full[lo:hi] + it).clone())
        g.send_view().copy_(full[lo:hi] * 2)  # a producer may write straight into the send buffer
        outs.append(g(g.send_view()).clone())
        q.put((rank, outs, ptrs == (g.send.data_ptr(), g.recv.data_ptr(), g.out.data_ptr())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 5])
def test_output_gatherer_reuses_its_buffers(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
    for _, outs, same in got:
        assert same
        for it in range(3):
            assert torch.equal(outs[it], full + it)
        assert torch.equal(outs[3], full * 2)
