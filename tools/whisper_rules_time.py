"""Cost of Whisper's logit filters in the decode step: 224-token greedy decode of 32 clips (Whisper-base), plain vs rules.
    python tools/whisper_rules_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models.audio2text import Whisper
from pytorch_models.audio2text.generate import WhisperRules
from synthweights import fill_module, synth_input, synth_tokens
torch.set_grad_enabled(False)
m = Whisper.from_openai("base").eval(); fill_module(m, 56); m = m.to(torch.bfloat16).cuda()
mem = m.encoder(synth_input("rt_mel", (32, 80, 3000), 1).cuda())
prompt = synth_tokens("rt_p", (32, 4), 50000, 2).cuda()
rules = WhisperRules(eot=50257, timestamp_begin=50364, no_timestamps=50363, max_initial_timestamp=50, suppress=tuple(range(1, 90)), blank=(220, 50257))
for r in (None, rules):
    m.decoder.generate(mem, prompt, 224, rules=r); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ids = m.decoder.generate(mem, prompt, 224, rules=r)
    torch.cuda.synchronize()
    print("rules" if r else "plain", f"{(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per 224-token decode of 32 clips")
