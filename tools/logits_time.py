"""Time of the step's vocabulary projection (final LayerNorm + logits + arg-max tiles: dec_logits_kernel) with E out of the caches,
as in a real step (0.8 GB of cross K/V streams between two uses).   python tools/logits_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import lib
L = lib()
torch.manual_seed(0)
spoil = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device="cuda")
for (M, K, V) in [(32, 512, 51865), (32, 384, 51865)]:
    x = torch.randn(M, K, device="cuda") * 3
    g = torch.rand(K, device="cuda") + 0.5
    b = torch.randn(K, device="cuda") * 0.1
    E = (torch.randn(V, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    tile = L.pm_dec_argmax_tile(K)
    nt = (V + tile - 1) // tile
    wv = torch.empty(M, nt, device="cuda")
    wi = torch.empty(M, nt, dtype=torch.int32, device="cuda")
    ts = []
    for it in range(16):
        spoil.fill_(it)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.pm_dec_linear(x.data_ptr(), K, g.data_ptr(), b.data_ptr(), 1e-5, E.data_ptr(), K, None, None, 0, None, 0, M, V, K, 0, 2,
                             None, None, 0, 0, 0, None, wv.data_ptr(), wi.data_ptr(), None)
        e1.record()
        assert rc == 0, rc
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print(f"M={M} K={K} V={V}: {t:6.2f} us  {V * K * 2 / t / 1e6:5.2f} TB/s", flush=True)
