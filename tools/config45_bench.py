"""Per-GPU rates of the two multi-GPU configurations of BASELINE.json on ONE MI355X (their shards are independent, so the
8-GPU figure is 8x this minus the all_gather of outputs): C5 = ViT-L/16 siglip @384, 256 images per GPU; C4 = Whisper
large-v2, 32 clips of 30 s per GPU, log-mel + encoder + 224-token greedy decode.  Synthetic weights and inputs.
    python tools/config45_bench.py [c5|c4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from synthweights import fill_module, synth_input, synth_tokens  # noqa: E402

torch.set_grad_enabled(False)
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
if which == "c5":
    from pytorch_models.image import ViT

    m = ViT.from_google("L/16_siglip", img_size=384)
    fill_module(m, 1)
    m = m.to(torch.bfloat16).cuda().eval()
    x = synth_input("c5_imgs", (256, 3, 384, 384), 2).cuda()
    m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        y = m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    gf = 383.9e9 * 256
    print(f"C5 ViT-L/16 siglip @384, 256 images: {dt * 1e3:.1f} ms  {256 / dt:.0f} img/s  model {gf / dt / 1e12:.0f} TFLOP/s ({gf / dt / 2.5e15:.3f} of peak)  out {tuple(y.shape)}")
else:
    from pytorch_models.audio2text import Whisper, WhisperPreprocessor

    m = Whisper.from_openai("large-v2").eval()
    fill_module(m, 1)
    m = m.to(torch.bfloat16).cuda()
    pre = WhisperPreprocessor("large-v2").cuda()
    wave = synth_input("c4_wave", (32, 480000), 2, scale=0.1).cuda()
    prompt = synth_tokens("c4_prompt", (32, 4), 51865, 3).cuda()
    ids = m.generate(pre(wave), prompt, 224)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        ids = m.generate(pre(wave), prompt, 224)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"C4 Whisper large-v2, 32 x 30 s clips, 224 greedy tokens: {dt * 1e3:.0f} ms  {32 * 30 / dt:.0f} audio-s/s  ids {tuple(ids.shape)}")
