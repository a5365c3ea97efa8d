"""Wav2Vec2 on MI355X: audio-seconds/s of the encoder forward on synthetic weights, with a
per-launch-group breakdown from HIP events (ops.LAUNCH_LOG).  Not a BASELINE metric - a measurement to go with the
parity tests of tests/test_hip_audio_enc.py.
    python tools/w2v_bench.py [--model base|large] [--batch 32] [--seconds 10]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402
from pytorch_models.audio import Wav2Vec2  # noqa: E402
from synthweights import bf16_round_, fill_module, synth_input  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="base")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
torch.set_grad_enabled(False)
if args.model == "base":  # facebook/wav2vec2-base, hubert-base: group-norm stem without conv bias, post-norm
    m = Wav2Vec2(12, 768, stem_bias=False, stem_legacy=True, pre_norm=False)
else:  # wav2vec2-large-lv60 / xls-r-300m: layer-norm stem with bias, pre-norm ("stable layer norm")
    m = Wav2Vec2(24, 1024, stem_bias=True, stem_legacy=False, pre_norm=True)
fill_module(m, 1)
bf16_round_(m)
m = m.to(torch.bfloat16).cuda().eval()
B, L = args.batch, int(args.seconds * 16000)
x = synth_input("w2v_bench", (B, L), 2).cuda()
y = m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(f"wav2vec2-{args.model} B={B} {args.seconds:g} s clips -> {tuple(y.shape)}: {dt * 1e3:8.2f} ms/step  "
      f"{B * args.seconds / dt:9.1f} audio-s/s", flush=True)
from pytorch_models.graph import GraphedForward  # noqa: E402

g = GraphedForward(m, x)
g(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    g(x)
torch.cuda.synchronize()
dtg = (time.perf_counter() - t0) / args.steps
print(f"  as one HIP graph (pytorch_models.graph.GraphedForward): {dtg * 1e3:8.2f} ms/step  {B * args.seconds / dtg:9.1f} audio-s/s", flush=True)
ops.LAUNCH_LOG = {}
m(x)
torch.cuda.synchronize()
log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
groups = {}
for name, evs in log.items():
    for a, b, work in evs:
        work = work[0] if isinstance(work, tuple) else work  # ops.linear logs (flop, bytes), the other wrappers a float
        g = groups.setdefault((name, work), [0, 0.0])
        g[0] += 1
        g[1] += a.elapsed_time(b)
tot = sum(v[1] for v in groups.values())
print(f"{'kernel':16s} {'work/launch':>12s} {'n':>4s} {'total ms':>9s} {'share':>6s} {'rate':>12s}")
for (name, work), (n, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    unit = "TFLOP/s" if name.startswith(("linear", "attention", "grouped_conv")) else "GB/s"
    rate = work * n / (ms * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
    print(f"{name:16s} {work:12.4g} {n:4d} {ms:9.3f} {ms / tot * 100:5.1f}% {rate:8.1f} {unit}")
print(f"sum of launches {tot:.2f} ms (event-timed, includes launch gaps inside a group)")
