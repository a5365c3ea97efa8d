"""Host-side profile of one eager Wav2Vec2 forward (where does the launch time go?)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models.audio import Wav2Vec2  # noqa: E402
from synthweights import fill_module, synth_input  # noqa: E402

torch.set_grad_enabled(False)
m = Wav2Vec2(12, 768, stem_bias=False, stem_legacy=True, pre_norm=False)
fill_module(m, 1)
m = m.to(torch.bfloat16).cuda().eval()
x = synth_input("w2v_bench", (2, 160000), 2).cuda()  # tiny batch: the GPU is never the bottleneck here
for _ in range(3):
    m(x)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    m(x)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
