"""Per-kernel view of one Whisper decode step (eager pass with HIP events, every 8th step sampled):
    python tools/whisper_decode_profile.py [tag=large-v2] [batch=32]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models.audio2text import Whisper  # noqa: E402
from pytorch_models.audio2text.generate import GreedyDecoder  # noqa: E402
from synthweights import fill_module, synth_input, synth_tokens  # noqa: E402

torch.set_grad_enabled(False)
tag = sys.argv[1] if len(sys.argv) > 1 else "large-v2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = Whisper.from_openai(tag).eval()
fill_module(m, 1)
m = m.to(torch.bfloat16).cuda()
d = m.decoder.token_embs.weight.shape[1]
memory = synth_input("wdp_mem", (B, 1500, d), 2).to(torch.bfloat16).cuda()
prompt = synth_tokens("wdp_p", (B, 4), 51865, 3).cuda()
dec = GreedyDecoder(m.decoder, memory, prompt, 224)
dec.reset()
log = {}
for i in range(dec.n_steps):
    dec.step(log if i % 8 == 0 else None)
torch.cuda.synchronize()
tot = 0.0
rows = []
for k, v in log.items():
    per = {}
    for a, b, args in v:
        key = (args[12], args[13], args[14]) if k == "pm_dec_linear" else (args[10], args[11]) if k == "pm_dec_linear_ksplit" else (
            "self" if args[18] else "cross",) if k.startswith("pm_dec_attention_fused") else ()
        per.setdefault(key, []).append(a.elapsed_time(b) * 1e3)
    nsteps = len(range(0, dec.n_steps, 8))
    for key, ts in per.items():
        per_step = sum(ts) / nsteps
        tot += per_step
        rows.append((per_step, f"{k:26s} {str(key):24s} launches/step {len(ts) / nsteps:5.1f}  avg {sum(ts) / len(ts):7.1f} us  per step {per_step:8.1f} us"))
for _, r in sorted(rows, reverse=True):
    print(r)
print(f"sum {tot:.0f} us per step (eager, event-timed)")
