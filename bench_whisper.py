"""Whisper leg of bench.py: log-mel + encoder + KV-cached greedy decode on 30 s synthetic clips."""
import os
import time

import torch

PEAK_HBM_GBS = 8000.0
N_NEW, PROMPT = 224, 4  # fixed prompt of 4 ids, max_seq_len // 2 new tokens
ROOT = os.path.dirname(os.path.abspath(__file__))


def _traffic(name):
    import json

    for r in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", r, name)
        if os.path.exists(path):
            return round(json.load(open(path))["traffic_bytes_per_launch"]), f"profiles/{r}/{name}"
    return None, None


def decode_step_bytes(B, d, n_layers, V, S, P, n_new, hid=None) -> float:
    """SURVEY.md 8(d), averaged over the steps of one run: every weight once (bf16), the packed cross K/V of every layer and
    sequence once, the self K/V written so far (position t reads t + 1 keys)."""
    hid = hid or 4 * d
    weights = 2 * ((8 * d * d + 2 * d * hid) * n_layers + V * d)
    cross = 2 * S * d * 2 * n_layers * B
    steps = P + n_new - 1
    self_kv = sum(2 * (t + 1) * d * 2 * n_layers * B for t in range(steps)) / steps
    return float(weights + cross + self_kv)


def cpu_baseline_whisper(cores: int) -> dict:
    """Oracle (kind "port") on the host, 2 clips of the same pipeline, both decode semantics:
    KV-cached (what the HIP path computes) and full-prefix recompute per token (what the reference's generators do)."""
    from oracle import ref_spectrogram as RS
    from oracle import ref_whisper as RW
    from pytorch_models.audio2text import Whisper
    from synthweights import fill_module, synth_input, synth_tokens

    torch.set_num_threads(cores)
    m = Whisper.from_openai("base")
    fill_module(m, 56)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    wave = synth_input("w_cpu", (2, 480000), 9, scale=0.1)
    prompt = synth_tokens("w_cpu_p", (2, PROMPT), 51865, 9)
    with torch.no_grad():
        t0 = time.perf_counter()
        mel = RS.whisper_log_mel(wave, 80, "rfft")
        mem = RW.encoder(sd, "encoder.", mel)
        t_front = time.perf_counter() - t0
        t0 = time.perf_counter()
        ids_c, _ = RW.greedy_cached(sd, "decoder.", prompt, mem, N_NEW)
        t_cached = time.perf_counter() - t0
        t0 = time.perf_counter()
        ids_r, _ = RW.greedy_recompute(sd, "decoder.", prompt, mem, N_NEW)
        t_rec = time.perf_counter() - t0
    same = bool(torch.equal(ids_c, ids_r))
    what = f"2 x 30 s clips: log-mel + 8-layer encoder ({t_front:.1f} s) + {N_NEW}-token greedy decode, fp32 oracle, torch threads={cores}"
    return {
        "cpu_baseline": {"value": round(60.0 / (t_front + t_cached), 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
                         "sample": f"{what}; KV-cached decode ({t_cached:.1f} s)"},
        "cpu_baseline_full_recompute": {"value": round(60.0 / (t_front + t_rec), 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
                                        "sample": f"{what}; full-prefix recompute per token, the reference's generator semantics "
                                                  f"({t_rec:.1f} s); ids equal to the cached loop's: {same}"},
    }


WHISPER_LEGS = {
    # model tag, per-GPU batch, metric name, BASELINE config
    "whisper": dict(tag="base", batch=32, name="Whisper-base", cfg="BASELINE configs[2]", traffic="whisper_step_traffic.json"),
    "c4": dict(tag="large-v2", batch=32, name="Whisper-large-v2", cfg="BASELINE configs[3]: batch 256 over 8 GPUs = 32 clips per GPU",
               traffic="c4_step_traffic.json"),
}


def exact_extra(m, pre, wave, prompt, fast_ids, ms_default):
    """Labelled extra of the Whisper-base leg: the reference-accuracy mode, generate(exact=True) - fp32 encoder on the fp32 twin of
    the bf16-valued weights, fp32 cross / self K/V in the graph-replayed step - timed once on the same clips.  Its ids are the
    reference's fp32 forward's on the same weights (tests/test_hip_exact.py pins that against the reference's own 224-token
    goldens, bit for bit); here: how much it costs and how far the default mode's ids are from it."""
    mel = pre(wave)
    ids = m.generate(mel, prompt, N_NEW, exact=True)  # builds the twin, captures the step graph
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids = m.generate(pre(wave), prompt, N_NEW, exact=True)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    B = wave.shape[0]
    again = m.generate(pre(wave), prompt, N_NEW, exact=True)
    return {"mode": "Whisper.generate(exact=True): fp32 encoder + fp32 K/V decode (same step kernels, one graph replay per token)",
            "value": round(B * 30.0 / (ms / 1e3), 1), "unit": "audio-s/s", "ms_per_step": round(ms, 2),
            "times_default_step": round(ms / ms_default, 2),
            "ids_equal_reference_semantics": "bit-exact vs the reference's fp32 greedy loop on the same bf16-valued weights: pinned by "
                                             "tests/test_hip_exact.py::test_exact_mode_of_the_bf16_model_equals_the_reference_bit_for_bit "
                                             "(tiny and base goldens, 2 x 228 ids each)",
            "rerun_identical": bool(torch.equal(ids, again)),
            "default_mode_agrees_on": round((fast_ids == ids).float().mean().item(), 4)}


def run_whisper(args, rank, world, device, sync, max_over_ranks, summarize_launches, timed_steps, leg="whisper"):
    from pytorch_models import dp
    from pytorch_models._hip import ops
    from pytorch_models.audio2text import Whisper, WhisperPreprocessor
    from pytorch_models.audio2text.generate import GreedyDecoder
    from synthweights import fill_module, synth_input, synth_tokens

    LEG = WHISPER_LEGS[leg]
    B = args.batch or LEG["batch"]
    tag = LEG["tag"]
    m = Whisper.from_openai(tag).eval()  # "base": 8 layers, exactly as the reference builds it
    n_layers = len(m.encoder.layers)
    if getattr(args, "whisper_layers", 0) and args.whisper_layers != n_layers:  # labelled extra, not the BASELINE config
        n_layers = args.whisper_layers
        m = Whisper(51865, n_layers, 512).eval()
    d_model = m.decoder.token_embs.weight.shape[1]
    fill_module(m, 56)
    m = m.to(torch.bfloat16).to(device)
    pre = WhisperPreprocessor(tag).to(device)
    wave = synth_input(f"w_bench_r{rank}", (B, 480000), 200 + rank, scale=0.1).to(device)
    prompt = synth_tokens(f"w_bench_p{rank}", (B, PROMPT), 51865, 200 + rank).to(device)
    gather = dp.OutputGatherer(B * world, (PROMPT + N_NEW,), torch.int64, device) if world > 1 else None
    path = getattr(args, "decode_path", "auto")

    with torch.no_grad():
        memory = m.encoder(pre(wave))
        dec = GreedyDecoder(m.decoder, memory, prompt, N_NEW, **({} if path == "auto" else {"path": path}))
        use_graph = not getattr(args, "no_graph", False)
        dec.run(graph=use_graph)  # builds + captures the step graph once

        def step():
            mem = m.encoder(pre(wave))
            dec.rebind(mem, prompt)
            toks = dec.run(graph=use_graph)
            return gather(toks) if gather is not None else toks

        dt = timed_steps(step, args, world)
        log = None
        t_decode = None
        if rank == 0:  # outside the timed region: per-kernel event timing of the eager (front end + encoder) launches ...
            from pytorch_models import transformer as _tf

            _tf.ENCODER_STREAMS = 1  # per-kernel durations need non-overlapping kernels (bench.py, run_vit)
            dec.rebind(m.encoder(pre(wave)), prompt)
            ops.LAUNCH_LOG = {}
            for _ in range(args.steps):
                dec.rebind(m.encoder(pre(wave)), prompt)
            torch.cuda.synchronize()
            log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
            _tf.ENCODER_STREAMS = 0
            # ... and the decode loop alone (the same graph replays as in the timed region) between two events
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(1, min(args.steps, 5))
            dec.run(graph=use_graph)
            e0.record()
            for _ in range(reps):
                dec.run(graph=use_graph)
            e1.record()
            torch.cuda.synchronize()
            t_decode = e0.elapsed_time(e1) / reps  # ms per 227-step decode
        sync(world)
    dt = max_over_ranks(dt, world, device)
    res = {
        "metric": f"{LEG['name']} audio-sec/s",
        "value": round(world * B * 30.0 * args.steps / dt, 1),
        "unit": "audio-s/s",
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "config": {"workload": f"{LEG['name']} ({('reference geometry: ' if leg == 'c4' or n_layers == 8 else 'EXTRA, not the BASELINE config: ') + str(n_layers)} layers, d={d_model}): log-mel + encoder + greedy decode "
                               f"(prompt {PROMPT}, {N_NEW} new tokens, KV cache), 30 s synthetic audio, batch={B} per GPU "
                               f"({LEG['cfg']})",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "decode_path": getattr(dec, "path", "launches"),
                   "collective": "dp.OutputGatherer: all_gather_into_tensor(token ids)" if world > 1 else "none"},
        "dtype": "bf16 weights and encoder activations, f32 decoder activations",
        "_dt": dt,
        "_layers": n_layers,
    }
    if rank == 0:
        kern = summarize_launches(log)
        res["kernels"] = {k: {"launches": v["n"], "total_ms": round(v["ms"], 3)} for k, v in kern.items()}
        S, d = memory.shape[1], memory.shape[2]
        V = m.decoder.token_embs.weight.shape[0]
        step_bytes = decode_step_bytes(B, d, n_layers, V, S, PROMPT, N_NEW)
        us_step = 1e3 * t_decode / dec.n_steps
        ach = step_bytes / us_step / 1e3  # bytes / us -> GB/s
        traffic, tfile = _traffic(LEG["traffic"])
        res["roofline"] = {
            "bound": "hbm", "kernel": f"whole decode step ({getattr(dec, 'path', 'launches')}: {len(dec.launches)} launches per step, one graph replay)",
            "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
            "traffic": traffic, "traffic_unit": f"bytes per decode step (FETCH_SIZE x2 + WRITE_SIZE over one eager step, rocprofv3 PMC, {tfile})",
            "algorithmic_bytes_per_step": round(step_bytes), "us_per_step": round(us_step, 2), "steps_per_decode": dec.n_steps,
            "decode_ms": round(t_decode, 3),
            "note": "HIP events around the graph replays of one whole decode (after the timed region); bytes = SURVEY.md 8(d) averaged over the run's positions"}
        # per-kernel view of one eager pass (HIP events on the launch stream): the dominant kernel of the step
        with torch.no_grad():
            dec.reset()
            dlog = {}
            for i in range(dec.n_steps):
                dec.step(dlog if i % 8 == 0 else None)  # sample every 8th step: keeps the event count bounded
            torch.cuda.synchronize()
        per = {}
        for k, v in dlog.items():
            n_per_step = sum(1 for f, _ in dec.launches if f.__name__ == k)
            per[k] = {"launches_per_step": n_per_step,
                      "us_per_step": round(1e3 * sum(a.elapsed_time(b) for a, b, _ in v) / (len(v) / n_per_step), 2)}
        res["decode_kernels_eager"] = per
        att = dlog.get("pm_dec_attention_fused_v2") or dlog.get("pm_dec_attention_fused", [])
        cross = [(a, b) for a, b, ar in att if ar[18] == 0]  # self_attn == 0 <=> cross-attention block
        if cross:
            cross_ms = sum(a.elapsed_time(b) for a, b in cross)
            cross_bytes = 2 * B * S * d * 2 + 2 * B * d * 4 + d * d * 2  # packed cross K/V (bf16) + x, out (f32) + the q weight once
            c_ach = len(cross) * cross_bytes / cross_ms / 1e6
            ctraffic, cfile = _traffic("whisper_traffic.json" if leg == "whisper" else "c4_traffic.json")
            res["roofline"]["dominant_kernel"] = {
                "kernel": "dec_attn_fused_kernel<false> (LN + q-proj + cross-attention over 1500 keys)", "achieved": round(c_ach, 1),
                "unit": "GB/s", "frac": round(c_ach / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": cross_bytes,
                "traffic": ctraffic, "traffic_file": cfile, "avg_launch_us": round(1e3 * cross_ms / len(cross), 2), "launches": len(cross)}
        if leg == "whisper" and n_layers == 8 and world == 1 and not getattr(args, "no_exact", False):
            with torch.no_grad():
                res["exact"] = exact_extra(m, pre, wave, prompt, dec.run(graph=use_graph).clone(), res["ms_per_step"])
    return res
