"""One attention shape, a few launches - the target of rocprofv3 --pmc passes.   python tools/attn_one.py B L H [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402

B, L, H = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
torch.manual_seed(0)
inner = H * 64
qkv = torch.randn(B, L, 3 * inner, device="cuda").to(torch.bfloat16)
for _ in range(iters):
    ops.attention(qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:], H, False, None)
torch.cuda.synchronize()
