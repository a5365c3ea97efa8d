"""Decode projections at Whisper large-v2's width (32 rows, d = 1280), weights of a different layer every launch (graph replay, no
launch gaps): the LayerNorm-fused kernels of the step against LayerNorm as its own launch + the plain kernels, and K splits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import ops
torch.manual_seed(0)
B, d, NW = 32, 1280, 32
x = torch.randn(B, d, device="cuda")
g = torch.ones(d, device="cuda"); be = torch.zeros(d, device="cuda")

def timed(fn_of_i):
    fn_of_i(0); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(NW):
            fn_of_i(i)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 / NW * 1e3

for name, N, K, act in (("qkv", 3840, 1280, "none"), ("fc1", 5120, 1280, "gelu"), ("out_proj", 1280, 1280, "none"), ("fc2", 1280, 5120, "none")):
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(NW)]
    b = torch.randn(N, device="cuda")
    xin = torch.randn(B, K, device="cuda")
    r = torch.randn(B, N, device="cuda")
    res = {}
    if K == d:
        res["LN fused"] = timed(lambda i: ops.dec_linear(xin, ws[i], b, ln=(g, be, 1e-5), act=act))
        res["LN launch + plain"] = timed(lambda i: ops.dec_linear(ops.layernorm(xin, g, be, 1e-5), ws[i], b, act=act))
    res["plain"] = timed(lambda i: ops.dec_linear(xin, ws[i], b, act=act, resid=r if act == "none" else None))
    for ks in (2, 4, 8):
        try:
            res[f"ksplit {ks}"] = timed(lambda i: ops.dec_linear_ksplit(xin, ws[i], b, k_split=ks, act=act, resid=r if act == "none" else None))
        except Exception as e:
            res[f"ksplit {ks}"] = float("nan")
    print(f"{name:9s} N={N:5d} K={K:5d}: " + " | ".join(f"{k} {v:6.2f}" for k, v in res.items()), flush=True)
