"""Per-kernel time of the fp32 (exact-mode) Whisper-base encoder on 32 clips: where the reference-accuracy mode's encoder spends."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import ops
from pytorch_models.audio2text import Whisper, WhisperPreprocessor
from synthweights import bf16_round_, fill_module, synth_input
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = Whisper.from_openai("base").eval()
fill_module(m, 56)
bf16_round_(m)
m = m.to(torch.bfloat16).cuda()
mel = WhisperPreprocessor("base").cuda()(synth_input("w_exact", (B, 480000), 9, scale=0.1).cuda())
twin = m.exact_copy()
twin.encoder(mel)
torch.cuda.synchronize()
ops.LAUNCH_LOG = {}
twin.encoder(mel)
torch.cuda.synchronize()
log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
tot = 0.0
for k, v in log.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in v)
    tot += ms
    print(f"{k:24s} launches {len(v):4d}  total {ms:8.2f} ms  avg {1e3 * ms / len(v):8.1f} us")
print(f"sum {tot:.1f} ms")
