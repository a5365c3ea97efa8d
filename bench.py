"""bench.py - the hot path on N MI355X of one node, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload both|vit|whisper|c4|c5|stub]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

BASELINE.json's metric has two halves and the default run times BOTH, each for exactly K steps after W warm-ups:
  vit     : ViT-B/16 bf16 forward, batch 256 per GPU, 224x224 (BASELINE.json configs[1])
  whisper : Whisper-base log-mel + encoder + 224-step KV-cached greedy decode, 32 x 30 s clips per GPU (configs[2])
  c4      : Whisper large-v2, the same pipeline, 32 clips per GPU = BASELINE configs[3] at --gpus 8 (batch 256 over 8 GPUs)
  c5      : ViT-L/16 siglip @384 bf16 forward, 256 images per GPU = BASELINE configs[4] at --gpus 8 (batch 2048 over 8 GPUs)
One "step" = one pass of the hot path over one per-GPU batch of synthetic input already resident in HBM.
Rank 0 prints ONE JSON line: the top-level metric fields are the ViT-B/16 leg (the configuration the metric is quoted on
first), the Whisper leg is the "whisper" object and is repeated under config.legs / roofline.whisper / cpu_baseline.whisper.
For N > 1 every rank runs its own shard (weak scaling, no data-path collective) and each step ends with the product's
one gather of the outputs (pytorch_models.dp.OutputGatherer: RCCL all_gather over xGMI under backend "nccl").
`--gpus N` without a torch.distributed environment starts the N ranks itself (child processes, before any HIP call).
`--workload stub` is a CPU stand-in step (no HIP) that drives exactly this file's N > 1 code path under gloo.
"""
import argparse
import json
import os
import socket
import subprocess
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
PROFILE_DIRS = ("r03", "r02", "r01")  # committed rocprofv3 PMC summaries, newest first


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
    """(HBM-side bytes per launch of the dominant kernel, file) from the committed rocprofv3 PMC passes
    (tools/collect_traffic.py; PMC cannot be sampled from inside this process).  (None, None) if nothing is committed."""
    for r in PROFILE_DIRS:
        path = os.path.join(ROOT, "profiles", r, name)
        if os.path.exists(path):
            return round(json.load(open(path))["traffic_bytes_per_launch"]), f"profiles/{r}/{name}"
    return None, None


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


def timed_steps(step, args, world):
    """W untimed warm-ups, then EXACTLY K steps between barrier + synchronize pairs; seconds of the timed region."""
    for _ in range(args.warmup):
        step()
    sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync(world)
    return time.perf_counter() - t0


VIT_LEGS = {
    # tag, image size, per-GPU batch, output width, FLOP per image, BASELINE config
    "vit": dict(tag="B/16", img=224, batch=256, d=768, flops=None, name="ViT-B/16", cfg="BASELINE configs[1]", traffic="vit_traffic.json"),
    "c5": dict(tag="L/16_siglip", img=384, batch=256, d=1024, flops=383.9e9, name="ViT-L/16 siglip @384",
               cfg="BASELINE configs[4]: batch 2048 over 8 GPUs = 256 per GPU", traffic="c5_traffic.json"),
}


def run_vit(args, rank, world, device, leg="vit"):
    from pytorch_models import dp
    from pytorch_models._hip import ops
    from pytorch_models.image import ViT
    from synthweights import fill_module, synth_input

    L = VIT_LEGS[leg]
    B = args.batch or L["batch"]
    m = (ViT.from_google(L["tag"]) if L["img"] == 224 else ViT.from_google(L["tag"], img_size=L["img"])).eval()
    fill_module(m, 32)
    m = m.to(torch.bfloat16).to(device)
    imgs = synth_input(f"{leg}_bench_r{rank}", (B, 3, L["img"], L["img"]), 100 + rank).to(device)
    gather = dp.OutputGatherer(B * world, (L["d"],), torch.bfloat16, device) if world > 1 else None
    flops_img = L["flops"] or vit_flops_per_image()

    def step():
        out = m(imgs)
        return gather(out) if gather is not None else out

    with torch.no_grad():
        dt = timed_steps(step, args, world)
        # roofline leg: the SAME K steps once more with two HIP events around every launch (recorded on the launch
        # stream).  Kept out of the timed region: the event records themselves open ~10 us gaps between kernels
        # (measured: 6 % of the step), which would tax `value` without changing the per-kernel durations.
        log = None
        if rank == 0:
            # ... and with the encoder on ONE stream: the product default runs the two halves of the batch on two streams
            # (pytorch_models/transformer.py, Encoder.forward), where the events of concurrent kernels overlap and a per-kernel
            # duration says nothing about the kernel.  Same kernels, same tile shapes, full-M launches.
            from pytorch_models import transformer as _tf

            _tf.ENCODER_STREAMS = 1
            for _ in range(2):
                m(imgs)
            ops.LAUNCH_LOG = {}
            for _ in range(args.steps):
                m(imgs)
            torch.cuda.synchronize()
            log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
            _tf.ENCODER_STREAMS = 0
        sync(world)
    dt = max_over_ranks(dt, world, device)
    res = {
        "metric": f"{L['name']} images/s",
        "value": round(world * B * args.steps / dt, 1),
        "unit": "images/s",
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "config": {"workload": f"{L['name']} bf16 forward, batch={B} per GPU, {L['img']}x{L['img']} ({L['cfg']})",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "collective": "dp.OutputGatherer: all_gather_into_tensor(outputs)" if world > 1 else "none"},
        "dtype": "bf16",
        "_dt": dt,
    }
    if rank == 0:
        kern = summarize_launches(log)
        lin = kern["linear_bf16"]
        ach = lin["work"] / lin["ms"] / 1e9  # flop / ms -> TFLOP/s
        traffic, tfile = measured_traffic(L["traffic"])
        n_layers = len(m.layers)
        lin_flops_step = lin["work"] / args.steps
        res["roofline"] = {"bound": "mfma", "kernel": f"linear_bf16 kernels (QKV, out_proj, fc1 + GELU, fc2 of {n_layers or 'all'} layers)",
                           "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                           "traffic_unit": f"bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 PMC, {tfile})",
                           "algorithmic_bytes_per_launch": round(lin["bytes"] / lin["n"]) if lin.get("bytes") else None,
                           "launches": lin["n"], "avg_launch_us": round(1e3 * lin["ms"] / lin["n"], 2),
                           # the timed region (product default: two halves of the batch on two streams) cannot be bracketed per
                           # kernel - concurrent kernels overlap -; what it does give: the linear kernels' flops over the WHOLE
                           # step time, a lower bound of their aggregate rate there (attention, patch embedding and the small
                           # kernels are inside that time too)
                           "default_schedule": {"streams": 2, "ms_per_step": round(1e3 * dt / args.steps, 3),
                                                "linear_tflop_per_step": round(lin_flops_step / 1e12, 3),
                                                "linear_tflops_over_whole_step": round(lin_flops_step / (dt / args.steps) / 1e12, 1),
                                                "frac_lower_bound": round(lin_flops_step / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4)},
                           "note": "achieved / avg_launch_us: per-kernel HIP-event timing of K extra steps with the encoder on ONE "
                                   "stream (full-M launches, no overlap; rocprofv3 --kernel-trace --stats of `PM_ENCODER_STREAMS=1 "
                                   f"python bench.py --workload {leg}` = profiles/r03/bench_{leg}_one_stream_kernel_stats.csv; of the "
                                   f"default two-stream run = profiles/r03/bench_{leg}_default_kernel_stats.csv)"}
        res["kernels"] = {k: {"launches": v["n"], "total_ms": round(v["ms"], 3)} for k, v in kern.items()}
        res["model_tflops"] = round(flops_img * B * args.steps / dt / 1e12, 1)
        res["model_frac_of_peak"] = round(res["model_tflops"] / PEAK_BF16_TFLOPS, 4)
    return res


def run_stub(args, rank, world, device):
    """CPU stand-in for a model step (tests/test_dp_gloo.py): per-sample arithmetic on this rank's shard, then the same
    gather, timing, max-over-ranks and JSON assembly as the real legs."""
    from pytorch_models import dp

    B = args.batch or 6
    x = torch.arange(B * 4, dtype=torch.float32).view(B, 4) + 1000.0 * rank
    gather = dp.OutputGatherer(B * world, (2,), torch.float32, device) if world > 1 else None
    last = {}

    def step():
        out = torch.stack([x.sum(1), x[:, 0] * 2], 1)
        last["out"] = gather(out) if gather is not None else out

    dt = timed_steps(step, args, world)
    dt = max_over_ranks(dt, world, device)
    want = torch.cat([torch.stack([(torch.arange(B * 4, dtype=torch.float32).view(B, 4) + 1000.0 * r).sum(1),
                                   (torch.arange(B * 4, dtype=torch.float32).view(B, 4) + 1000.0 * r)[:, 0] * 2], 1)
                      for r in range(world)])
    return {"metric": "stub samples/s", "value": round(world * B * args.steps / dt, 1), "unit": "samples/s",
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "config": {"workload": "stub (CPU stand-in step, no HIP)", "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": f"dp{world}", "collective": "dp.OutputGatherer" if world > 1 else "none"},
            "dtype": "f32", "gather_ok": bool(torch.equal(last["out"], want)), "_dt": dt}


def summarize_launches(log):
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    out = {}
    for name, evs in (log or {}).items():
        out[name] = {"n": len(evs), "ms": sum(a.elapsed_time(b) for a, b, _ in evs),
                     "work": sum((w[0] if isinstance(w, tuple) else w) for _, _, w in evs),
                     "bytes": sum((w[1] if isinstance(w, tuple) else 0) for _, _, w in evs)}
    return out


def sync(world):
    if world > 1:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(dt, world, device):
    if world == 1:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def spawn_ranks(args) -> int:
    """--gpus N > 1 without RANK / WORLD_SIZE: start the N ranks as children of this process (nothing here has touched
    the GPU yet) and relay rank 0's JSON line.  Never an exec: see the environment notes on exec after HIP init."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="both", choices=["both", "vit", "whisper", "c4", "c5", "stub"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (0 = the BASELINE config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="whisper: skip the labelled exact-mode extra (generate(exact=True))")
    ap.add_argument("--whisper-layers", type=int, default=0,
                    help="whisper: layers per stack (0 = 8, the reference's \"base\"; 6 = OpenAI's base geometry, a labelled extra: SURVEY.md F2)")
    ap.add_argument("--backend", default="nccl", help='torch.distributed backend ("nccl" = RCCL; "gloo" only to rehearse N > 1 on one GPU or on CPU)')
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-graph", action="store_true", help="whisper: eager decode launches instead of the captured HIP graph "
                    "(rocprofv3 --pmc cannot sample graph replays on this stack)")
    ap.add_argument("--decode-path", default="auto", choices=["auto", "launches", "persistent"],
                    help="whisper: force the per-stage launch list or the persistent layer kernel of the decode step")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         "(or without a torch.distributed environment: bench.py then starts the ranks itself)")
    cpu_only = args.workload == "stub"
    if args.single_device:
        local = 0
    if world > 1:  # before any HIP call
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if cpu_only:
        device = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        if world > 1:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm
            else:
                dist.init_process_group(args.backend)

    legs = {}
    if args.workload == "stub":
        legs["stub"] = run_stub(args, rank, world, device)
    if args.workload in ("both", "vit"):
        legs["vit"] = run_vit(args, rank, world, device)
    if args.workload == "c5":
        legs["c5"] = run_vit(args, rank, world, device, "c5")
    if args.workload in ("both", "whisper", "c4"):
        from bench_whisper import run_whisper

        wleg = "c4" if args.workload == "c4" else "whisper"
        legs[wleg] = run_whisper(args, rank, world, device, sync, max_over_ranks, summarize_launches, timed_steps, leg=wleg)

    if rank == 0:
        want_cpu = not args.no_cpu_baseline and world == 1  # rank 0, N = 1 only
        if want_cpu and "vit" in legs:
            legs["vit"]["cpu_baseline"] = cpu_baseline_vit()
        if want_cpu and "whisper" in legs and legs["whisper"]["_layers"] == 8 and "large" not in legs["whisper"]["metric"]:  # the CPU leg is the BASELINE geometry's
            from bench_whisper import cpu_baseline_whisper

            legs["whisper"].update(cpu_baseline_whisper(host_cores()))
        for leg in legs.values():
            leg.pop("_dt", None)
            leg.pop("_layers", None)
        first = next(iter(legs.values()))
        res = dict(first)
        if "whisper" in legs and "vit" in legs:
            w = legs["whisper"]
            res["metric"] = ("ViT-B/16 images/s & Whisper-base audio-sec/s (BASELINE.json metric; value / unit / ms_per_step = the "
                             "ViT-B/16 leg of configs[1], the Whisper-base leg of configs[2] is the \"whisper\" object)")
            res["config"] = dict(first["config"], legs={k: {"workload": v["config"]["workload"], "value": v["value"], "unit": v["unit"],
                                                             "ms_per_step": v["ms_per_step"]} for k, v in legs.items()})
            res["roofline"] = dict(first["roofline"], whisper=w.get("roofline"))
            if "cpu_baseline" in first:
                res["cpu_baseline"] = dict(first["cpu_baseline"], whisper=w.get("cpu_baseline"),
                                           whisper_full_recompute=w.get("cpu_baseline_full_recompute"))
            res["whisper"] = w
        res.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
                    "vs_baseline": None, "data": "synthetic"})
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
