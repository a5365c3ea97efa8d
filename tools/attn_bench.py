"""Micro-benchmark of pm_attention_bf16 on the ViT-B/16 (B=256, L=197, H=12) and Whisper-base encoder (B=32, L=1500, H=8)
self-attention shapes, reading the packed q/k/v projection in place.    python tools/attn_bench.py [--iters 30]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)  # a window of tens of ms: see tools/_timing.py
args = ap.parse_args()
torch.manual_seed(0)
for name, B, L, H, causal in (("vit-b/16", 256, 197, 12, False), ("whisper-base enc", 32, 1500, 8, False), ("causal 448", 32, 448, 8, True),
                              ("vit-l/16 @384 half batch", 128, 576, 16, False)):
    inner = H * 64
    qkv = torch.randn(B, L, 3 * inner, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:]
    for _ in range(100):
        ops.attention(q, k, v, H, causal, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.attention(q, k, v, H, causal, None)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    fl = 4.0 * B * H * L * L * 64 * (0.5 if causal else 1.0)
    print(f"{name:26s} B={B} L={L} H={H}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
