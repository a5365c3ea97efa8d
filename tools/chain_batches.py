"""Chained decode step (PM_DEC_CHAIN=1, the default) against the launch list without deferred sums over batch sizes 1 .. 32 of
Whisper-base: same ids, reruns identical.   python tools/chain_batches.py"""
import os, sys, torch
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch-models_amd"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
from synthweights import bf16_round_, fill_module, synth_input, synth_tokens
from pytorch_models.audio2text import Whisper
torch.set_grad_enabled(False)
w = Whisper.from_openai("base").eval(); fill_module(w, 56); bf16_round_(w); w = w.to(torch.bfloat16).cuda()
for B in (1, 3, 5, 17, 31, 32):
    mem = synth_input("cb_mem", (B, 1500, 512), 7 + B).to(torch.bfloat16).cuda()
    prompt = synth_tokens("cb_p", (B, 4), 51865, B).cuda()
    os.environ["PM_DEC_CHAIN"] = "1"; a = w.decoder.generate(mem, prompt, 48)
    a2 = w.decoder.generate(mem, prompt, 48)
    os.environ["PM_DEC_CHAIN"] = "0"; b = w.decoder.generate(mem, prompt, 48)
    print(B, "chain == plain:", bool(torch.equal(a, b)), "rerun identical:", bool(torch.equal(a, a2)), "mismatch positions:", int((a != b).sum()))
