"""The phase-interleaved 256x256x64 K loop (csrc/linear_bf16_8ph.hip, experiment) against the product's wide kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd"), os.path.join(ROOT, "tools")]
import torch
from pytorch_models._hip import lib, ops
from _timing import time_us
L = lib()
f = L.pm_gemm8ph_bench
f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
f.restype = ctypes.c_int
torch.manual_seed(0)
for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192), (50432, 2304, 768), (50432, 3072, 768), (50432, 768, 3072)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    assert f(x.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), N, M, N, K, st) == 0
    torch.cuda.synchronize()
    ref = ops.linear(x, w, None)
    rows = slice(0, M, 61)
    ref32 = x[rows].float() @ w.float().T
    err = (y[rows].float() - ref32).abs().max().item()
    same = (y == ref).float().mean().item()
    t = time_us
    t8 = t(lambda: f(x.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), N, M, N, K, st))
    tw = t(lambda: ops.linear(x, w, None, out=ref))
    print(f"M={M} N={N} K={K}: 8-phase {t8:7.1f} us {2*M*N*K/t8/1e6:7.1f} TF | product {tw:7.1f} us {2*M*N*K/tw/1e6:7.1f} TF | max|err| {err:.3e} equal-to-product {same:.4f}", flush=True)
