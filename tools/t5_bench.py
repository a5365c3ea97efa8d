"""T5 v1.1 on MI355X: teacher-forced tokens/s of the full model (encoder over S source tokens + decoder over L target tokens +
fp32 logits) on synthetic weights, with the per-kernel breakdown of ops.LAUNCH_LOG.  Not a BASELINE metric - a measurement to
go with tests/test_hip_t5.py.     python tools/t5_bench.py [--size base] [--batch 32] [--src 512] [--tgt 128]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402
from pytorch_models.text import T5Model  # noqa: E402
from synthweights import bf16_round_, fill_module, synth_tokens  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="base")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--src", type=int, default=512)
ap.add_argument("--tgt", type=int, default=128)
args = ap.parse_args()
torch.set_grad_enabled(False)
m = T5Model.from_t5x(f"t5_1_1-{args.size}")
fill_module(m, 1)
bf16_round_(m)
m = m.to(torch.bfloat16).cuda().eval()
B, S, L = args.batch, args.src, args.tgt
src = synth_tokens("t5_bench_src", (B, S), 32128, 2).cuda()
tgt = synth_tokens("t5_bench_tgt", (B, L), 32128, 3).cuda()
m(src, tgt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m(src, tgt)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"t5_1_1-{args.size} B={B} src={S} tgt={L}: {dt * 1e3:8.2f} ms/step  {B * (S + L) / dt / 1e3:9.1f} k tokens/s (source + target)", flush=True)
ops.LAUNCH_LOG = {}
m(src, tgt)
torch.cuda.synchronize()
log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
tot = sum(a.elapsed_time(b) for evs in log.values() for a, b, _ in evs)
for name, evs in sorted(log.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b, _ in kv[1])):
    ms = sum(a.elapsed_time(b) for a, b, _ in evs)
    print(f"{name:16s} {len(evs):4d} launches {ms:8.3f} ms {ms / tot * 100:5.1f}%")
