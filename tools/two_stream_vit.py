"""Does ViT-B/16 at batch 256 run faster as two half batches on two HIP streams (the tail rounds of one half's persistent
kernels filled by the other half's workgroups)?   python tools/two_stream_vit.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models.image import ViT
from synthweights import fill_module, synth_input

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = ViT.from_google("B/16").eval()
fill_module(m, 32)
m = m.to(torch.bfloat16).cuda()
imgs = synth_input("vit_bench_r0", (B, 3, 224, 224), 100).cuda()
side = torch.cuda.Stream()

def plain():
    return m(imgs)

def split(parts):
    def run():
        cur = torch.cuda.current_stream()
        n = B // parts
        outs = [None] * parts
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for i in range(1, parts, 2):
                outs[i] = m(imgs[i * n:(i + 1) * n])
        for i in range(0, parts, 2):
            outs[i] = m(imgs[i * n:(i + 1) * n])
        cur.wait_stream(side)
        return torch.cat(outs)
    return run

def bench(fn, steps=20, warm=5):
    with torch.no_grad():
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

with torch.no_grad():
    ref = plain()
    for name, fn in (("plain", plain), ("2 halves / 2 streams", split(2)), ("4 quarters / 2 streams", split(4)), ("plain", plain), ("2 halves / 2 streams", split(2))):
        ms = bench(fn)
        same = torch.equal(fn(), ref)
        print(f"{name:26s}: {ms:7.3f} ms/step  {B / ms:7.2f} k img/s  identical to plain: {same}", flush=True)
