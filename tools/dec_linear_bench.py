"""How long is a decode projection when its weights are L2-hot (same weights replayed) vs cold (a different layer's
weights every launch, 64 MB of them: the decode step's situation)?  Graph replay, so no launch gaps.
    python tools/dec_linear_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402

torch.manual_seed(0)
B = 32
for name, N, K, ks in (("out_proj", 512, 512, 0), ("fc1+ln+gelu", 2048, 512, 0), ("fc2 plain", 512, 2048, 0)):
    NW = 64
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(NW)]
    x = torch.randn(B, K, device="cuda")
    b = torch.randn(N, device="cuda")
    r = torch.randn(B, N, device="cuda")
    g = torch.ones(K, device="cuda")
    junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")

    def call(w):
        if name.startswith("fc1"):
            return ops.dec_linear(x, w, b, ln=(g, g, 1e-5), act="gelu")
        return ops.dec_linear(x, w, b, resid=r)

    res = {}
    for mode in ("hot", "cold"):
        call(ws[0])
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(NW):
                call(ws[0] if mode == "hot" else ws[i])
                if mode == "cold":
                    pass
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            if mode == "cold":
                junk.fill_(1)  # flush L2 / MALL between replays (timed too: subtract below)
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        junk.fill_(1)
    e1.record()
    torch.cuda.synchronize()
    fill = e0.elapsed_time(e1) / 10
    print(f"{name:14s} N={N} K={K}: hot {res['hot'] / NW * 1e3:6.2f} us/launch   cold {(res['cold'] - fill) / NW * 1e3:6.2f} us/launch", flush=True)
