"""The (64 MI) x 256 tile kernels (csrc/linear_bf16_tile.hip) against an fp32 reference and the other kernels: values, row
partials, LayerNorm-fold consumer, edges (M not a multiple of the tile, N not a multiple of 256), determinism; then timings.
    PM_GEMM_KERNEL=6|7 python tools/tile_check.py [--time]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd"), os.path.join(ROOT, "tools")]
import torch
from pytorch_models._hip import ops
from _timing import time_us
torch.manual_seed(0)
which = os.environ.get("PM_GEMM_KERNEL", "auto")
bad = 0
cases = [(50432, 768, 768, "none", True, True, False), (50432, 768, 3072, "none", True, True, False),
         (50432, 3072, 768, "gelu", False, False, True), (50432, 2304, 768, "none", False, False, True),
         (4104, 520, 192, "none", True, True, False), (4104, 520, 192, "gelu", False, False, False),
         (8200, 1000, 128, "none", False, False, True), (25216, 768, 768, "none", True, False, False),
         (4096, 256, 64, "none", False, False, False)]
for (M, N, K, act, resid, rows, lnc) in ([] if "--time-only" in sys.argv else cases):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if resid else None
    rows = rows and N % 64 == 0
    kw = {}
    if lnc:
        st = torch.stack([torch.randn(M, device="cuda") * 0.1, torch.rand(M, device="cuda") + 0.5], 1).contiguous()
        s = w.float().sum(1).contiguous()
        kw = dict(ln_stats=st, ln_s=s)
    out = ops.linear(x, w, b, act=act, resid=r, want_row_stats=rows, **kw)
    out, part = out if rows else (out, None)
    out2 = ops.linear(x, w, b, act=act, resid=r, want_row_stats=rows, **kw)
    out2 = out2[0] if rows else out2
    sel = torch.cat([torch.arange(0, M, 97, device="cuda"), torch.arange(max(0, M - 40), M, device="cuda")])
    ref = x[sel].float() @ w.float().T
    if lnc:
        ref = st[sel, 1:2] * (ref - st[sel, 0:1] * s) + b
    else:
        ref = ref + b
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    if resid:
        ref = ref + r[sel].float()
    err = (out[sel].float() - ref).abs().max().item()
    tol = 2 ** -7 * ref.abs().max().item() + 1e-2
    ok = err <= tol and torch.equal(out, out2) and bool(torch.isfinite(out.float()).all())
    msg = ""
    if rows:
        blk = out.float().view(M, N // 64, 64)
        e1 = (part[..., 0] - blk.sum(-1)).abs().max().item()
        e2 = ((part[..., 1] - (blk * blk).sum(-1)).abs() / (1 + (blk * blk).sum(-1))).max().item()
        ok = ok and e1 < 1e-3 and e2 < 1e-4
        msg = f" rowstats err {e1:.2e} {e2:.2e}"
    bad += not ok
    print(f"kernel={which} M={M} N={N} K={K} act={act} resid={resid} rows={rows} lnc={lnc}: max|err| {err:.3e} (tol {tol:.3e}) rerun-identical {bool(torch.equal(out, out2))}{msg} {'OK' if ok else 'FAIL'}", flush=True)
if "--time" in sys.argv or "--time-only" in sys.argv:
    for (M, N, K, act, resid, rows, lnc) in cases[:4] + [(25216, 768, 768, "none", True, True, False), (25216, 768, 3072, "none", True, True, False),
                                                      (25216, 3072, 768, "gelu", False, False, True), (25216, 2304, 768, "none", False, False, True),
                                                      (8192, 8192, 8192, "none", False, False, False)]:
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if resid else None
        kw = {}
        if lnc:
            st = torch.stack([torch.randn(M, device="cuda") * 0.1, torch.rand(M, device="cuda") + 0.5], 1).contiguous()
            kw = dict(ln_stats=st, ln_s=w.float().sum(1).contiguous())
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        us = time_us(lambda: ops.linear(x, w, b, act=act, resid=r, out=out, want_row_stats=rows, **kw))
        print(f"time kernel={which} M={M} N={N} K={K} act={act} resid={resid} rows={rows} lnc={lnc}: {us:7.1f} us {2*M*N*K/us/1e6:7.1f} TF", flush=True)
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
