"""Which of the three GEMM kernels wins where: pm_linear_bf16 over transformer-block shapes, one line per shape.  Run once per
forced kernel (PM_GEMM_KERNEL = 1: 128 x 128, 2: 256 x 128 persistent, 3: 256 x 256 persistent, each wherever it applies) and
join the outputs - the cost model of linear_impl (pm_linear_pick_kernel) was fitted to this table."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402

torch.set_grad_enabled(False)
RESID = "--resid" in sys.argv  # the residual GEMMs (out_proj, linear2: N = d_model) with their bf16 residual operand
Ms = [4096, 6144, 8192, 12288, 15968, 25216, 32768, 50432]
NKs = [(768, 768), (768, 3072), (2304, 768), (3072, 768), (512, 512), (512, 2048), (1536, 512), (2048, 512), (1024, 1024),
       (1024, 4096), (3072, 1024), (4096, 1024)]
if RESID:
    NKs = [(n, k) for n, k in NKs if n <= 1024]
for N, K in NKs:
    for M in Ms:
        r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if RESID else None
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K**0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        act = "gelu" if N > K else "none"
        for _ in range(3):
            ops.linear(x, w, b, act=act, resid=r)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.linear(x, w, b, act=act, resid=r)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"{M} {N} {K} {us:.1f} {2.0 * M * N * K / us / 1e6:.0f}", flush=True)
