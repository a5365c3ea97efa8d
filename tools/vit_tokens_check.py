"""pm_vit_tokens: SHA-256 of the output for a few geometries (compare the two tilings: PM_VIT_TOKENS_IMAGE=0 / 1), or its time.
    python tools/vit_tokens_check.py            -> one line per geometry: shape, sha256
    python tools/vit_tokens_check.py --time     -> us per launch and GB/s of algorithmic bytes at ViT-B/16 b = 256 / 128 and ViT-L/16 @384"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402
from synthweights import synth_input  # noqa: E402

P = 16


def make(N, H, W, d, cls):
    L = (H // P) * (W // P)
    imgs = synth_input("vc_img", (N, 3, H, W), 50).cuda()
    w = synth_input("vc_w", (d, 3 * P * P), 51, scale=0.04).to(torch.bfloat16).cuda()
    b = synth_input("vc_b", (d,), 52, scale=0.1).cuda()
    pe = synth_input("vc_pe", (L, d), 53, scale=0.1).cuda()
    c = synth_input("vc_cls", (d,), 54, scale=0.1).cuda() if cls else None
    return imgs, w, b, pe, c


if "--time" in sys.argv:
    for (N, H, W, d, cls) in [(256, 224, 224, 768, True), (128, 224, 224, 768, True), (256, 384, 384, 1024, False)]:
        a = make(N, H, W, d, cls)
        L = (H // P) * (W // P)
        ts = []
        for it in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.vit_tokens(*a, P)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 5)
        t = sorted(ts)[len(ts) // 2]
        nbytes = N * 3 * H * W * 4 + N * (L + int(cls)) * d * 2
        print(f"N={N} {H}x{W} d={d}: {t:7.1f} us  {nbytes / t / 1e3:6.0f} GB/s of algorithmic bytes", flush=True)
else:
    for (N, H, W, d, cls) in [(5, 224, 224, 768, True), (9, 224, 224, 256, False), (2, 384, 384, 1024, False), (3, 240, 208, 512, True),
                              (8, 224, 224, 768, True), (1, 16, 16, 256, True)]:
        out = ops.vit_tokens(*make(N, H, W, d, cls), P)
        torch.cuda.synchronize()
        print((N, H, W, d, cls), tuple(out.shape), hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest(), flush=True)
