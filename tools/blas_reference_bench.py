"""How fast does the vendor library (hipBLASLt through torch.nn.functional.linear) run the ViT-B/16 GEMM shapes on this
box?  A yardstick for pm_linear_bf16 only - nothing in the product calls it.
    python tools/blas_reference_bench.py [--iters 30]"""
import argparse

import torch
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=150)  # tens of milliseconds per window: see tools/_timing.py
ap.add_argument("--zeros", action="store_true")
args = ap.parse_args()
SHAPES = [(50432, 2304, 768), (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072), (8192, 8192, 8192), (48000, 1536, 512)]
torch.manual_seed(0)
for M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    if args.zeros:
        x.zero_(), w.zero_()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(100):
        F.linear(x, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        F.linear(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"hipBLASLt M={M:6d} N={N:5d} K={K:5d}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
