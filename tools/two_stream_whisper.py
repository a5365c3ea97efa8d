"""Whisper-base encoder (32 x 1500 tokens) with the Encoder on one stream and on two."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models import transformer as tf
from pytorch_models._hip import ops
from pytorch_models.audio2text import Whisper
from synthweights import fill_module, synth_input

m = Whisper.from_openai("base").eval()
fill_module(m, 56)
m = m.to(torch.bfloat16).cuda()
mel = synth_input("w_mel", (32, 80, 3000), 3).cuda()

def bench(steps=20, warm=5):
    with torch.no_grad():
        for _ in range(warm):
            m.encoder(mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.encoder(mel)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

for mode in (1, 2, 1, 2):
    tf.ENCODER_STREAMS = mode
    ms = bench()
    ops.LAUNCH_LOG = {}
    with torch.no_grad():
        m.encoder(mel)
    torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    per = {k: (len(v), round(sum(a.elapsed_time(b) for a, b, *_ in v), 3)) for k, v in log.items()}
    print(f"streams={mode}: {ms:7.3f} ms per encoder forward; launches {per}", flush=True)

# ---- the bench step in sections (encoder, cross K/V projection = rebind, 227 graph replays)
from pytorch_models.audio2text import WhisperPreprocessor
from pytorch_models.audio2text.generate import GreedyDecoder
from synthweights import synth_tokens
pre = WhisperPreprocessor("base").cuda()
wave = synth_input("w_bench_r0", (32, 480000), 200, scale=0.1).cuda()
prompt = synth_tokens("w_bench_p0", (32, 4), 51865, 200).cuda()
with torch.no_grad():
    memory = m.encoder(pre(wave))
    dec = GreedyDecoder(m.decoder, memory, prompt, 224)
    dec.run(graph=True)
    for mode in (1, 2, 1, 2):
        tf.ENCODER_STREAMS = mode
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        tot = [0.0, 0.0, 0.0]
        for it in range(8):
            ev[0].record()
            mem = m.encoder(pre(wave))
            ev[1].record()
            dec.rebind(mem, prompt)
            ev[2].record()
            dec.run(graph=True)
            ev[3].record()
            torch.cuda.synchronize()
            if it >= 3:
                for j in range(3):
                    tot[j] += ev[j].elapsed_time(ev[j + 1]) / 5
        print(f"streams={mode}: front end + encoder {tot[0]:7.3f} ms | rebind {tot[1]:7.3f} ms | decode {tot[2]:7.3f} ms", flush=True)

# ---- the same step without a host synchronisation per step (as bench.py times it)
with torch.no_grad():
    for mode in (1, 2, 1, 2):
        tf.ENCODER_STREAMS = mode
        for sync_each in (False, True):
            for _ in range(3):
                dec.rebind(m.encoder(pre(wave)), prompt); dec.run(graph=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                dec.rebind(m.encoder(pre(wave)), prompt); dec.run(graph=True)
                if sync_each:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            print(f"streams={mode} sync_each_step={sync_each}: {(time.perf_counter() - t0) / 8 * 1e3:8.3f} ms per step", flush=True)
