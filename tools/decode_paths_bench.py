"""us per decode step (graph replay, HIP events) of the two step forms, same weights and memory:
    python tools/decode_paths_bench.py [tag=base] [batch=32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models.audio2text import Whisper
from pytorch_models.audio2text.generate import GreedyDecoder
from synthweights import fill_module, synth_input, synth_tokens
torch.set_grad_enabled(False)
tag = sys.argv[1] if len(sys.argv) > 1 else "base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m = Whisper.from_openai(tag).eval(); fill_module(m, 1); m = m.to(torch.bfloat16).cuda()
d = m.decoder.token_embs.weight.shape[1]
memory = synth_input("dpb_mem", (B, 1500, d), 2).to(torch.bfloat16).cuda()
prompt = synth_tokens("dpb_p", (B, 4), 51865, 3).cuda()
res = {}
for path in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("launches", "persistent")):
    dec = GreedyDecoder(m.decoder, memory, prompt, 224, path=path)
    dec.run(); dec.check()
    toks = dec.tokens.clone()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dec.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / dec.n_steps)
    dec.check()
    res[path] = (min(ts), toks)
    print(f"{tag} B={B} {path}: {min(ts):.1f} us/step ({len(dec.launches)} launches per step)", flush=True)
ks = list(res)
print("ids equal:", all(bool(torch.equal(res[ks[0]][1], res[k][1])) for k in ks))
