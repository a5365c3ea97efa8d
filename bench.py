"""bench.py - the hot path on N MI355X of one node, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload vit|whisper]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one per-GPU batch of synthetic input already resident in HBM:
  vit     : ViT-B/16 bf16 forward, batch 256 per GPU, 224x224 (BASELINE.json configs[1])
  whisper : Whisper-base log-mel + encoder + 224-step greedy decode, 32 x 30 s clips per GPU (configs[2])
For N > 1 every rank runs its own shard (weak scaling, no data-path collective) and the step ends with
the one RCCL all_gather of the outputs.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pytorch-models_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM_GBS = 8000.0


def host_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 64)


def measured_traffic(name: str):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (tools/collect_traffic.py;
    PMC cannot be sampled from inside this process).  None if no measurement is committed."""
    path = os.path.join(ROOT, "profiles", "r01", name)
    if not os.path.exists(path):
        return None
    return round(json.load(open(path))["traffic_bytes_per_launch"])


def vit_flops_per_image(n_layers=12, d=768, L=197, patches=196, k_patch=768) -> float:
    """n_layers * (24 L d^2 + 4 L^2 d) + 2 * patches * K * d."""
    return n_layers * (24 * L * d * d + 4 * L * L * d) + 2 * patches * k_patch * d


def cpu_baseline_vit(seconds_budget: float = 20.0) -> dict:
    """The oracle (kind "port") timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import ref_vit
    from pytorch_models.image import ViT
    from synthweights import fill_module, synth_input

    cores = host_cores()
    torch.set_num_threads(cores)
    m = ViT.from_google("B/16")
    fill_module(m, 32)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    geo = ref_vit.geometry_from_google("B/16")
    x = synth_input("vit_cpu", (8, 3, 224, 224), 7)
    with torch.no_grad():
        ref_vit.forward(sd, geo, x[:2])  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            ref_vit.forward(sd, geo, x)
            n += x.shape[0]
            if time.perf_counter() - t0 > seconds_budget or n >= 64:
                break
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} images (batches of 8) of ViT-B/16 224x224, fp32 oracle, torch threads={cores}"}


def run_vit(args, rank, world, device):
    from pytorch_models._hip import ops
    from pytorch_models.image import ViT
    from synthweights import fill_module, synth_input

    B = args.batch or 256
    m = ViT.from_google("B/16").eval()
    fill_module(m, 32)
    m = m.to(torch.bfloat16).to(device)
    imgs = synth_input(f"vit_bench_r{rank}", (B, 3, 224, 224), 100 + rank).to(device)
    gathered = [torch.empty(B, 768, dtype=torch.bfloat16, device=device) for _ in range(world)] if world > 1 else None

    def step():
        out = m(imgs)
        if world > 1:
            dist.all_gather(gathered, out)
        return out

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        sync(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync(world)
        dt = time.perf_counter() - t0
        # roofline leg: the SAME K steps once more with two HIP events around every launch (recorded on the launch
        # stream).  Kept out of the timed region: the event records themselves open ~10 us gaps between kernels
        # (measured: 6 % of the step), which would tax `value` without changing the per-kernel durations.
        log = None
        if rank == 0:
            ops.LAUNCH_LOG = {}
            for _ in range(args.steps):
                m(imgs)
            torch.cuda.synchronize()
            log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
        sync(world)
    dt = max_over_ranks(dt, world, device)
    res = {
        "metric": "ViT-B/16 images/s (BASELINE.json: Whisper-base audio-sec/s & ViT-B/16 images/s)",
        "value": round(world * B * args.steps / dt, 1),
        "unit": "images/s",
        "config": {"workload": f"ViT-B/16 bf16 forward, batch={B} per GPU, 224x224 (BASELINE configs[1])",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "collective": "all_gather(outputs)" if world > 1 else "none"},
        "dtype": "bf16",
        "_dt": dt,
    }
    if rank == 0:
        kern = summarize_launches(log)
        lin = kern["linear_bf16"]
        ach = lin["work"] / lin["ms"] / 1e9  # flop / ms -> TFLOP/s
        res["roofline"] = {"bound": "mfma", "kernel": "linear_bf16_kernel", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                           "traffic": measured_traffic("vit_traffic.json"), "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 PMC, profiles/r01/vit_traffic.json)",
                           "launches": lin["n"], "avg_launch_us": round(1e3 * lin["ms"] / lin["n"], 2)}
        res["kernels"] = {k: {"launches": v["n"], "total_ms": round(v["ms"], 3)} for k, v in kern.items()}
        res["model_tflops"] = round(vit_flops_per_image() * B * args.steps / dt / 1e12, 1)
        res["model_frac_of_peak"] = round(res["model_tflops"] / PEAK_BF16_TFLOPS, 4)
    return res


def summarize_launches(log):
    torch.cuda.synchronize()
    out = {}
    for name, evs in (log or {}).items():
        out[name] = {"n": len(evs), "ms": sum(a.elapsed_time(b) for a, b, _ in evs), "work": sum(w for _, _, w in evs)}
    return out


def sync(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(dt, world, device):
    if world == 1:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="vit", choices=["vit", "whisper"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (0 = the BASELINE config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--whisper-layers", type=int, default=0,
                    help="whisper: layers per stack (0 = 8, the reference's \"base\"; 6 = OpenAI's base geometry, a labelled extra: SURVEY.md F2)")
    ap.add_argument("--backend", default="nccl", help='torch.distributed backend ("nccl" = RCCL; "gloo" only to rehearse N > 1 on one GPU)')
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-graph", action="store_true", help="whisper: eager decode launches instead of the captured HIP graph "
                    "(rocprofv3 --pmc cannot sample graph replays on this stack)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    if args.workload == "vit":
        res = run_vit(args, rank, world, device)
    else:
        from bench_whisper import run_whisper

        res = run_whisper(args, rank, world, device, sync, max_over_ranks, summarize_launches, host_cores)

    if rank == 0:
        dt = res.pop("_dt")
        res.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
        if not args.no_cpu_baseline and world == 1:  # rank 0, N = 1 only
            if args.workload == "vit":
                res["cpu_baseline"] = cpu_baseline_vit()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
