"""Stream-K GEMM against the whole-tile kernels on the same operands: bit-level agreement is not expected (a split tile sums
its two K ranges separately), closeness to an fp32 reference is; plus timings of the ViT-B/16 shapes with each kernel."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd"), os.path.join(ROOT, "tools")]
import torch
from pytorch_models._hip import ops
from _timing import time_us
torch.manual_seed(0)
which = os.environ.get("PM_GEMM_KERNEL", "auto")
for (M, N, K, act, resid) in [(50432, 768, 768, "none", True), (50432, 768, 3072, "none", True), (50432, 3072, 768, "gelu", False),
                              (50432, 2304, 768, "none", False), (8192, 8192, 8192, "none", False), (20000, 1024, 512, "none", True)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if resid else None
    out = ops.linear(x, w, b, act=act, resid=r)
    out2 = ops.linear(x, w, b, act=act, resid=r)
    rows = slice(0, M, 97)
    ref = x[rows].float() @ w.float().T + b
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if resid:
        ref = ref + r[rows].float()
    err = (out[rows].float() - ref).abs().max().item()
    rel = ((out[rows].float() - ref).norm() / ref.norm()).item()
    us = time_us(lambda: ops.linear(x, w, b, act=act, resid=r, out=out))
    print(f"kernel={which} M={M} N={N} K={K} act={act} resid={resid}: {us:7.1f} us {2*M*N*K/us/1e6:7.1f} TF  max|err| {err:.3e} rel {rel:.2e} rerun-identical {bool(torch.equal(out, out2))}", flush=True)
