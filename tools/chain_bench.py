"""Micro-benchmark of the decode step's fused attention blocks as links of the deferred-sum chain (pm_dec_attention_chain):
the plain block, + IN (n parts added while the row loads), + OUT (per-head partial sums of the output projection)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import check, lib  # noqa: E402

L = lib()
B, H, d, S, T, pos = int(os.environ.get("CB_B", "32")), int(os.environ.get("CB_H", "8")), int(os.environ.get("CB_D", "512")), 1500, 228, 120
inner = H * 64
torch.manual_seed(0)
dev = "cuda"
g, be = torch.ones(d, device=dev), torch.zeros(d, device=dev)
x, x2 = torch.randn(B, d, device=dev), torch.empty(B, d, device=dev)
wq = (torch.randn(3 * inner, d, device=dev) / d ** 0.5).to(torch.bfloat16)
bq = torch.randn(3 * inner, device=dev)
wo = (torch.randn(d, inner, device=dev) / d ** 0.5).to(torch.bfloat16)
bo = torch.randn(d, device=dev)
kv = torch.randn(B, S, 2 * inner, device=dev).to(torch.bfloat16)
kc, vc = torch.randn(B, H, T, 64, device=dev).to(torch.bfloat16), torch.randn(B, H, T, 64, device=dev).to(torch.bfloat16)
posv = torch.tensor([pos], dtype=torch.int32, device=dev)
parts = torch.randn(8, B, d, device=dev)
hp = torch.empty(B, H, d, device=dev)
att = torch.empty(B, inner, device=dev)
st = None


def run(self_attn, n_in, emit, chain=True):
    if self_attn:
        a = (wq.data_ptr(), bq.data_ptr(), kc.data_ptr(), vc.data_ptr(), H * T * 64, T * 64, 64, posv.data_ptr(), 0, T)
    else:
        a = (wq.data_ptr(), bq.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 2, S * 2 * inner, 64, 2 * inner, None, S, S)
    if not chain:
        return L.pm_dec_attention_fused(x.data_ptr(), d, g.data_ptr(), be.data_ptr(), 1e-5, *a, att.data_ptr(), B, H, self_attn, st)
    return L.pm_dec_attention_chain(x.data_ptr(), d, g.data_ptr(), be.data_ptr(), 1e-5, *a, B, H, self_attn, 0,
                                    parts.data_ptr() if n_in else None, n_in, B * d, d, bo.data_ptr() if n_in else None,
                                    x2.data_ptr() if n_in else None, wo.data_ptr() if emit else None, hp.data_ptr() if emit else None,
                                    None if emit else att.data_ptr(), st)


spoil = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device=dev)
for self_attn in (1, 0):
    for name, kw in (("plain (pm_dec_attention_fused)", dict(n_in=0, emit=0, chain=False)), ("chain, nothing deferred", dict(n_in=0, emit=0)),
                     ("IN 4 parts", dict(n_in=4, emit=0)), ("IN 8 parts", dict(n_in=8, emit=0)), ("OUT head partials", dict(n_in=0, emit=1)),
                     ("IN 4 + OUT", dict(n_in=4, emit=1))):
        ts = []
        for it in range(14):
            if not self_attn:
                spoil.fill_(it)  # the cross K/V never sits in the Infinity Cache in a real step
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4 if self_attn else 1):
                check(run(self_attn, **kw), "launch")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / (4 if self_attn else 1))
        print(f"{'self ' if self_attn else 'cross'} {name:34s} {sorted(ts)[len(ts) // 2]:7.2f} us", flush=True)
