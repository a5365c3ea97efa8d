"""Cost of the reference-accuracy mode: Whisper-base, 32 x 30 s clips, 224 greedy tokens - default (bf16 encoder, bf16 K/V) against
generate(exact=True) (fp32 encoder on the fp32 twin, fp32 K/V in the graph-replayed step), split into encoder and decode.
    python tools/exact_time.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models.audio2text import Whisper, WhisperPreprocessor
from pytorch_models.audio2text.generate import GreedyDecoder
from synthweights import bf16_round_, fill_module, synth_input, synth_tokens
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = Whisper.from_openai("base").eval()
fill_module(m, 56)
bf16_round_(m)
m = m.to(torch.bfloat16).cuda()
pre = WhisperPreprocessor("base").cuda()
wave = synth_input("w_exact", (B, 480000), 9, scale=0.1).cuda()
prompt = synth_tokens("w_exact_p", (B, 4), 51865, 9).cuda()
mel = pre(wave)


def t(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms_d, ids_d = t(lambda: m.generate(mel, prompt, 224))
ms_e, ids_e = t(lambda: m.generate(mel, prompt, 224, exact=True))
twin = m.exact_copy()
ms_enc32, mem32 = t(lambda: twin.encoder(mel))
ms_enc16, _ = t(lambda: m.encoder(mel))
dec = GreedyDecoder(m.decoder, mem32, prompt, 224, kv32=True)
ms_dec32, _ = t(lambda: dec.run())
print(f"B={B}: default {ms_d:.1f} ms, exact {ms_e:.1f} ms ({ms_e / ms_d:.2f}x); encoder bf16 {ms_enc16:.1f} / fp32 {ms_enc32:.1f} ms; "
      f"decode with fp32 K/V {ms_dec32:.1f} ms ({1e3 * ms_dec32 / dec.n_steps:.0f} us per step); ids agree with default on "
      f"{(ids_d == ids_e).float().mean().item():.3f}")
