"""First contact of the persistent decode kernel with the GPU: tiny geometry, bounded by the caller's timeout."""
import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/pytorch-models_amd"]
import torch
from synthweights import bf16_round_, fill_module, synth_input, synth_tokens
from pytorch_models.audio2text import Whisper
from pytorch_models.audio2text.generate import GreedyDecoder
torch.set_grad_enabled(False)
d, L, B, S = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (128, 2, 2, 50)
w = Whisper(1000, L, d).eval(); fill_module(w, 3); w = w.to(torch.bfloat16).cuda()
mem = synth_input("m", (B, S, d), 1).to(torch.bfloat16).cuda()
prompt = synth_tokens("p", (B, 3), 1000, 2)
ps = GreedyDecoder(w.decoder, mem, prompt.cuda(), 9, path="persistent")
ln = GreedyDecoder(w.decoder, mem, prompt.cuda(), 9, path="launches")
ps.reset(); ln.reset()
for i in range(ps.n_steps):
    ps.step(); ln.step(); torch.cuda.synchronize()
    print(i, "err", int(ps.err.item()), "max|dx|", float((ps.x - ln.x).abs().max()), flush=True)
print("tokens equal:", bool(torch.equal(ps.tokens, ln.tokens)))
