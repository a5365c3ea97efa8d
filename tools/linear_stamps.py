"""Phase stamps of dec_linear_kernel (a -DPM_DF_STAMPS=1 build via PM_MI355X_LIB) for the three projections of a Whisper-base decode
step: out_proj (32 x 512 x 512 + bias + residual), fc1 (LayerNorm + GELU, N = 2048), fc2 as K parts (K = 2048, 4 parts).
Median / max over workgroups of the time since the earliest workgroup start (100 MHz clock: 10 ns steps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import lib
L = lib()
L.pm_debug_df_stamps.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
dev = "cuda"
M, d, hid = 32, 512, 2048
x = torch.randn(M, d, device=dev)
h = torch.randn(M, hid, device=dev)
g, be = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev) * 0.1
wo = (torch.randn(d, d, device=dev) / d ** 0.5).to(torch.bfloat16)
w1 = (torch.randn(hid, d, device=dev) / d ** 0.5).to(torch.bfloat16)
w2 = (torch.randn(d, hid, device=dev) / hid ** 0.5).to(torch.bfloat16)
bo, b1 = torch.randn(d, device=dev), torch.randn(hid, device=dev)
out, hout, parts = torch.empty(M, d, device=dev), torch.empty(M, hid, device=dev), torch.empty(4, M, d, device=dev)
junk = torch.empty(64 * 1024 * 1024, dtype=torch.uint8, device=dev)
names = ["start", "loads issued", "x here, LayerNorm done", "split + MFMA", "partials in LDS, barrier", "summed, epilogue, stored"]


def show(label, fn, nwg):
    for it in range(5):
        junk.fill_(it)  # the operands leave the L2s like between two launches of a real step
        assert fn() == 0
        torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * (1024 * 16))()
    assert L.pm_debug_df_stamps(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.int64).view(1024, 16)
    t = t[t[:, 0] > t[:, 0].max() - 10000]  # the slots this launch wrote (within 100 us of its latest start): older launches' stay behind
    t0 = t[:, 0].min()
    print(f"{label}: {t.shape[0]} workgroups, starts spread over {(t[:, 0].max() - t0) * 0.01:.2f} us")
    for i in range(1, 6):
        r = (t[:, i] - t0).double() * 0.01
        print(f"   {names[i]:28s} median {r.median():6.2f} us   max {r.max():6.2f}")


show("out_proj", lambda: L.pm_dec_linear(x.data_ptr(), d, None, None, 0.0, wo.data_ptr(), d, bo.data_ptr(), out.data_ptr(), d, out.data_ptr(), d,
                                          M, d, d, 0, 0, None, None, 0, 0, 0, None, None, None, None), 64)
show("fc1 (LayerNorm + GELU)", lambda: L.pm_dec_linear(x.data_ptr(), d, g.data_ptr(), be.data_ptr(), 1e-5, w1.data_ptr(), d, b1.data_ptr(), None, 0,
                                                        hout.data_ptr(), hid, M, hid, d, 1, 0, None, None, 0, 0, 0, None, None, None, None), 256)
show("fc2 as 4 K parts", lambda: L.pm_dec_linear_kparts(h.data_ptr(), hid, w2.data_ptr(), hid, parts.data_ptr(), d, M * d, M, d, hid, 4, None), 256)
