"""Micro-benchmark of pm_linear_bf16 on the ViT-B/16 shapes (and square references), random data.
    python tools/gemm_bench.py [--shapes vit|square|all] [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="all")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--act", default="none")
ap.add_argument("--resid", action="store_true")
ap.add_argument("--pad", type=int, default=0, help="extra elements in the row pitch of x and w (L2 channel spread experiment)")
ap.add_argument("--zeros", action="store_true", help="all-zero operands: the clock the chip holds on data that toggles nothing")
args = ap.parse_args()
VIT = [(50432, 2304, 768), (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072)]
SQ = [(4096, 4096, 4096), (8192, 8192, 8192), (48000, 1024, 1024), (48000, 1536, 512), (48000, 512, 2048)]
shapes = dict(vit=VIT, square=SQ, all=VIT + SQ)[args.shapes]
torch.manual_seed(0)
for M, N, K in shapes:
    x = torch.randn(M, K + args.pad, device="cuda").to(torch.bfloat16)[:, :K]
    w = (torch.randn(N, K + args.pad, device="cuda") / K ** 0.5).to(torch.bfloat16)[:, :K]
    b = torch.randn(N, device="cuda")
    if args.zeros:
        x.zero_(), w.zero_()
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if args.resid else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(x, w, b, act=args.act, resid=r, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.linear(x, w, b, act=args.act, resid=r, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"M={M:6d} N={N:5d} K={K:5d} act={args.act} resid={args.resid}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
