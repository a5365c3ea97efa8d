"""One grouped positional conv of wav2vec2-base size (32 x 499 steps, d = 768, 16 groups, k = 128) for the PMC passes of
tools/pmc_gconv.sh: 3 launches of pm_group_windows + pm_grouped_conv_bf16."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models.audio import Wav2Vec2  # noqa: E402

torch.set_grad_enabled(False)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 768
conv = torch.nn.Conv1d(d, d, 128, groups=16).to(torch.bfloat16).cuda()
h = torch.randn(32, 499, d, device="cuda").to(torch.bfloat16)
for _ in range(3):
    y = Wav2Vec2.grouped_conv(conv, h, (64, 63), "gelu", h)
torch.cuda.synchronize()
print(tuple(y.shape))
