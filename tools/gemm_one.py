"""One GEMM shape, a few launches - the target of rocprofv3 --pmc passes.   python tools/gemm_one.py M N K [iters] [act]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
act = sys.argv[5] if len(sys.argv) > 5 else "none"
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(iters):
    ops.linear(x, w, b, act=act, out=out)
torch.cuda.synchronize()
