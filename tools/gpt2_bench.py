"""GPT-2 small on MI355X: teacher-forced forward tokens/s and KV-cached greedy decode tokens/s,
synthetic weights.  Not a BASELINE metric - a measurement to go with the parity tests of tests/test_hip_text.py.
    python tools/gpt2_bench.py [--batch 32] [--prompt 64] [--new 192]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models.text import GPT2  # noqa: E402
from synthweights import bf16_round_, fill_module, synth_tokens  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--prompt", type=int, default=64)
ap.add_argument("--new", type=int, default=192)
args = ap.parse_args()
torch.set_grad_enabled(False)
m = GPT2.from_hf("gpt2")
fill_module(m, 1)
bf16_round_(m)
m = m.to(torch.bfloat16).cuda().eval()
B, P, N = args.batch, args.prompt, args.new
tok = synth_tokens("gpt2_bench", (B, 1024), 50257, 2).cuda()
for L in (256, 1024):
    x = tok[:, :L]
    m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"forward  B={B} L={L}: {dt * 1e3:8.2f} ms  {B * L / dt / 1e3:8.1f} k tokens/s (logits over 50257 included)", flush=True)
prompt = tok[:, :P].contiguous()
m.generate(prompt, N)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    m.generate(prompt, N)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"generate B={B} prompt={P} new={N}: {dt * 1e3:8.2f} ms  {B * N / dt / 1e3:8.2f} k new tokens/s  ({dt / (P + N - 1) * 1e6:.0f} us per step)", flush=True)
