"""Epilogue stamps of the large-M GEMM (a -DPM_TILE_STAMPS=1 build via PM_MI355X_LIB): per workgroup's FIRST tile, the time from
the first K step to the last K step's barrier, to the epilogue's start, and behind every 16-token block of wave 0 (median over
the 256 workgroups, us; 100 MHz clock: 10 ns steps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import lib, ops
L = lib()
L.pm_debug_tile_stamps.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
M = 50432
for (N, K, act, res) in [(768, 768, "none", True), (768, 3072, "none", True), (3072, 768, "gelu", False), (2304, 768, "none", False)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    kw = {}
    if not res:
        kw = dict(ln_stats=torch.stack([torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")], 1).contiguous(), ln_s=torch.zeros(N, device="cuda"))
    for _ in range(3):
        y = ops.linear(x, w, b, act=act, resid=r, want_row_stats=res, **kw)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * (256 * 16))()
    assert L.pm_debug_tile_stamps(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.int64).view(256, 16)
    rel = lambda a, b_: ((t[:, a] - t[:, b_]).double() * 0.01).median().item()
    nb = 10 if os.environ.get("PM_GEMM_KERNEL", "7") == "7" else 8
    print(f"N={N} K={K} act={act} resid={res}: K loop (first to last step) {rel(0, 14):6.2f} us | last step's barrier -> epilogue start {rel(1, 0):5.2f} | "
          + " ".join(f"{rel(2 + q, 1 + q):4.2f}" for q in range(nb)) + f" | epilogue {rel(1 + nb, 1):6.2f} us")
