"""Host time to ENQUEUE one ViT-B/16 forward at batch 256 (two-stream schedule) against the GPU time it takes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models.image import ViT
from synthweights import fill_module, synth_input
m = ViT.from_google("B/16").eval(); fill_module(m, 32); m = m.to(torch.bfloat16).cuda()
x = synth_input("vit_bench_r0", (256, 3, 224, 224), 100).cuda()
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m(x)
    t_host = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 3
print(f"host enqueue {t_host * 1e3:.2f} ms per forward (GPU-bound pacing included if the queue fills); wall {t_all * 1e3:.2f} ms per forward")
