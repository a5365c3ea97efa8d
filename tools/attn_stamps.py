"""Phase stamps of the short-sequence attention kernel (a -DPM_AH_STAMPS=1 build via PM_MI355X_LIB): per wave, one head in the
steady state of its workgroup (ViT-B/16: B = 256, H = 12, L = 197), median over the 256 workgroups, us (100 MHz clock)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import lib, ops
L = lib()
L.pm_debug_ah_stamps.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
B, Lq, H = int(os.environ.get("AS_B", 256)), int(os.environ.get("AS_L", 197)), 12
qkv = torch.randn(B, Lq, 3 * H * 64, device="cuda").to(torch.bfloat16)
q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
for _ in range(20):
    ops.attention(q, k, v, H, False, None)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * (256 * 8 * 8))()
assert L.pm_debug_ah_stamps(buf) == 0
t = torch.tensor(list(buf), dtype=torch.int64).view(256, 8, 8).double() * 0.01
names = ["wait own loads", "barrier", "issue next + store prev", "QK^T", "max", "exp + PV", "scale + stage"]
print("phase (us, median over workgroups)   " + " ".join(f"wave{w:2d}" for w in range(8)))
for i, n in enumerate(names):
    print(f"{n:36s} " + " ".join(f"{(t[:, w, i + 1] - t[:, w, i]).median().item():6.2f}" for w in range(8)))
print(f"{'head, stamp 0 to 7':36s} " + " ".join(f"{(t[:, w, 7] - t[:, w, 0]).median().item():6.2f}" for w in range(8)))
t0 = t[:, :, 0].min(dim=1, keepdim=True).values
print("spread of the waves' loop tops within a workgroup (us):", f"{(t[:, :7, 0] - t0).max(dim=1).values.median().item():.2f}")
