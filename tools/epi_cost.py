"""What the GEMM epilogue's operands cost: the same shape with / without bias, residual, GELU (kernel forced by PM_GEMM_KERNEL)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd"), os.path.join(ROOT, "tools")]
import torch
from pytorch_models._hip import ops
from _timing import time_us
torch.manual_seed(0)
t = time_us
for (M, N, K) in [(50432, 768, 3072), (50432, 3072, 768), (50432, 2304, 768), (50432, 768, 768)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = []
    for (name, kw) in [("plain", {}), ("bias", {"b": b}), ("resid", {"resid": r}), ("bias+resid", {"b": b, "resid": r}),
                       ("bias+gelu", {"b": b, "act": "gelu"}), ("plain", {})]:
        bb = kw.get("b")
        res.append(f"{name} {t(lambda: ops.linear(x, w, bb, act=kw.get('act', 'none'), resid=kw.get('resid'), out=out)):6.1f}")
    print(f"M={M} N={N} K={K}: " + " | ".join(res), flush=True)
