"""Phase stamps of dec_logits_kernel (a -DPM_DF_STAMPS=1 build via PM_MI355X_LIB): per stamp the median / max over workgroups of
the time since the EARLIEST workgroup start (100 MHz clock)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import lib
L = lib()
L.pm_debug_df_stamps.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
spoil = torch.empty(600 * 1024 * 1024, dtype=torch.uint8, device="cuda")
M, K, V = 32, 512, 51865
x = torch.randn(M, K, device="cuda") * 3
g = torch.rand(K, device="cuda") + 0.5
b = torch.randn(K, device="cuda") * 0.1
E = (torch.randn(V, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
tile = L.pm_dec_argmax_tile(K)
nt = (V + tile - 1) // tile
wv = torch.empty(M, nt, device="cuda")
wi = torch.empty(M, nt, dtype=torch.int32, device="cuda")
for cold in (1, 0):
    for it in range(4):
        if cold:
            spoil.fill_(it)
        rc = L.pm_dec_linear(x.data_ptr(), K, g.data_ptr(), b.data_ptr(), 1e-5, E.data_ptr(), K, None, None, 0, None, 0, M, V, K, 0, 2,
                             None, None, 0, 0, 0, None, wv.data_ptr(), wi.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * (1024 * 16))()
    assert L.pm_debug_df_stamps(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.int64).view(1024, 16)[:512]
    t0 = t[:, 0].min()
    print("E cold (HBM)" if cold else "E warm (Infinity Cache)", f": workgroup starts spread over {(t[:, 0].max() - t0) * 0.01:.2f} us")
    for i in range(1, 8):
        col = t[:, i]
        ok = col >= t0
        if ok.sum() == 0:
            continue
        r = (col[ok] - t0).double() * 0.01
        print(f"   stamp {i}: {int(ok.sum()):4d} workgroups  median {r.median():6.2f} us  max {r.max():6.2f}")
