"""Per-kernel view of one GPT-2 small decode step (eager pass with HIP events, every 8th step sampled).
    python tools/gpt2_decode_profile.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models.text import GPT2
from pytorch_models.audio2text.generate import GreedyDecoder
from synthweights import bf16_round_, fill_module, synth_tokens
torch.set_grad_enabled(False)
m = GPT2.from_hf("gpt2"); fill_module(m, 1); bf16_round_(m); m = m.to(torch.bfloat16).cuda().eval()
tok = synth_tokens("g", (32, 64), 50257, 2).cuda()
dec = GreedyDecoder(m, None, tok, 192)
dec.reset()
log = {}
for i in range(dec.n_steps):
    dec.step(log if i % 8 == 0 else None)
torch.cuda.synchronize()
for k, v in log.items():
    per = {}
    for a, b, args in v:
        key = (args[12], args[13], args[14]) if k == "pm_dec_linear" else (args[10], args[11]) if k == "pm_dec_linear_ksplit" else (args[18],) if k.startswith("pm_dec_attention_fused") else ()
        per.setdefault(key, []).append(a.elapsed_time(b) * 1e3)
    for key, ts in per.items():
        print(f"{k:26s} {str(key):22s} n={len(ts):4d} avg {sum(ts)/len(ts):7.1f} us")
