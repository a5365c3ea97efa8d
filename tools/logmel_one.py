"""A few launches of the Whisper log-mel front end on 32 x 30 s clips - the target of rocprofv3 --pmc passes
(tools/hbm_kernels_traffic.sh).     python tools/logmel_one.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models.audio2text import WhisperPreprocessor  # noqa: E402
from synthweights import synth_input  # noqa: E402

torch.set_grad_enabled(False)
pre = WhisperPreprocessor("base").cuda()
wave = synth_input("lm_wave", (32, 480000), 1, scale=0.1).cuda()
for _ in range(4):
    mel = pre(wave)
torch.cuda.synchronize()
print(tuple(mel.shape))
