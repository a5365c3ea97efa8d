"""Cost of the LayerNorm fold per ViT-B/16 GEMM: plain vs consumer (ln_stats) vs producer (row partials).
    python tools/lnfold_bench.py [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--M", type=int, default=50432)
args = ap.parse_args()
M = args.M
CASES = [("qkv", 2304, 768, "none", False, "consume"), ("out_proj", 768, 768, "none", True, "produce"),
         ("fc1", 3072, 768, "gelu", False, "consume"), ("fc2", 768, 3072, "none", True, "produce")]
torch.manual_seed(0)


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


for name, N, K, act, resid, mode in CASES:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if resid else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    stats = torch.rand(M, 2, device="cuda") + 0.5
    s = torch.randn(N, device="cuda")
    plain = timeit(lambda: ops.linear(x, w, b, act=act, resid=r, out=out))
    if mode == "consume":
        fold = timeit(lambda: ops.linear(x, w, b, act=act, resid=r, out=out, ln_stats=stats, ln_s=s))
    else:
        fold = timeit(lambda: ops.linear(x, w, b, act=act, resid=r, out=out, want_row_stats=True))
    fl = 2 * M * N * K / 1e6
    print(f"{name:9s} M={M} N={N:5d} K={K:5d}: plain {plain:7.1f} us ({fl/plain:6.1f} TF)   {mode} {fold:7.1f} us ({fl/fold:6.1f} TF)   "
          f"delta {fold-plain:+6.1f} us", flush=True)
x = torch.randn(M, 768, device="cuda").to(torch.bfloat16)
g = torch.ones(768, device="cuda")
print(f"layernorm M={M} d=768: {timeit(lambda: ops.layernorm(x, g, g, 1e-6)):7.1f} us", flush=True)
rows = torch.rand(M, 12, 2, device="cuda")
print(f"ln_stats_finalize: {timeit(lambda: ops.ln_stats_finalize(rows, 768, 1e-6)):7.1f} us", flush=True)
