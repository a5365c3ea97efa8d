"""Micro-benchmark of the fused cross-attention decode block over batch size (workgroups = B * 8)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch  # noqa: E402

from pytorch_models._hip import check, lib  # noqa: E402

L = lib()
d, H, S = int(os.environ.get("CB_D", "512")), int(os.environ.get("CB_H", "8")), 1500
torch.manual_seed(0)
g = torch.ones(d, device="cuda")
be = torch.zeros(d, device="cuda")
w = (torch.randn(d, d, device="cuda") / d ** 0.5).to(torch.bfloat16)
LAYOUT = sys.argv[1] if len(sys.argv) > 1 else "token"
for B in ([int(os.environ["CB_B"])] if "CB_B" in os.environ else (8, 16, 24, 32, 48, 64)):
    x = torch.randn(B, d, device="cuda")
    kv = torch.randn(B, S, 2 * d, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, d, device="cuda")
    spoil = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    times = []
    for it in range(12):
        if not os.environ.get("PM_BENCH_WARM"):
            spoil.fill_(it)  # evict the K/V from the Infinity Cache like the other layers' streams do (PM_BENCH_WARM=1: keep it)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if LAYOUT == "token":  # (B, S, [k | v]) as the projection GEMM writes it: a head's keys are 128 B every 2 KiB
            rc = L.pm_dec_attention_fused(x.data_ptr(), d, g.data_ptr(), be.data_ptr(), 1e-5, w.data_ptr(), None, kv.data_ptr(),
                                          kv.data_ptr() + d * 2, S * 2 * d, 64, 2 * d, None, S, S, out.data_ptr(), B, H, 0, st)
        else:  # (B, 2H, S, 64): every (sequence, head) stream is one contiguous 192 KiB run
            rc = L.pm_dec_attention_fused(x.data_ptr(), d, g.data_ptr(), be.data_ptr(), 1e-5, w.data_ptr(), None, kv.data_ptr(),
                                          kv.data_ptr() + H * S * 64 * 2, 2 * H * S * 64, S * 64, 64, None, S, S,
                                          out.data_ptr(), B, H, 0, st)
        e1.record()
        check(rc, "fused")
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(times)[len(times) // 2]
    print(f"B={B:3d} WGs={B*H:4d}: {t:7.1f} us  {2*B*S*d*2/t/1e3:7.1f} GB/s", flush=True)
