"""Phase stamps of the fused cross / self attention block (a -DPM_DF_STAMPS=1 build: PM_MI355X_LIB=.../var_stamps/libpm_mi355x.so).
Prints, per phase, the median over workgroups of the time since the workgroup's first stamp (100 MHz wall clock: 10 ns steps)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402

import chain_bench as cb  # noqa: E402  (runs its table once, then we reuse its buffers)

L = cb.L
L.pm_debug_df_stamps.argtypes = [ctypes.c_void_p]
names = ["start", "x + parts requested .. weights, K requested", "LayerNorm", "q projection", "K pass", "V requested, softmax",
         "(P.V start)", "P.V", "tail: reduce / out", "OUT partials"]
for self_attn, kw, label in ((0, dict(n_in=0, emit=0, chain=False), "cross plain"), (0, dict(n_in=8, emit=0), "cross IN 8"),
                             (1, dict(n_in=0, emit=0, chain=False), "self plain"), (1, dict(n_in=4, emit=1), "self IN 4 + OUT")):
    rows = []
    for it in range(6):
        if not self_attn:
            cb.spoil.fill_(it)
        cb.check(cb.run(self_attn, **kw), "launch")
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * (1024 * 16))()
        assert L.pm_debug_df_stamps(buf) == 0
        t = torch.tensor(list(buf), dtype=torch.int64).view(1024, 16)[: min(cb.B * cb.H, 1024)]
        rows.append(t)
    t = rows[-1]
    n = 9 if kw.get("emit") else 8
    rel = (t[:, :n] - t[:, :1]).double() * 0.01  # us
    med = rel.median(0).values
    first = (t[:, 0] - t[:, 0].min()).double() * 0.01
    srt = first.sort().values
    print("   workgroup start times (us), every 32nd:", [round(float(v), 1) for v in srt[::32]])
    hw = t[:, 15]
    cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)  # xcc, se, sh, cu
    print("   placement (xcc*1000 + se*100 + sh*50 + cu) of workgroups 0..15:", cu[:16].tolist(), " 256..263:", cu[256:264].tolist(),
          " distinct CUs:", len(set(cu.tolist())))
    from collections import Counter
    print("   workgroups per CU:", sorted(Counter(Counter(cu.tolist()).values()).items()))
    print(f"{label}: workgroup starts spread over {first.max():.2f} us; last end {((t[:, n - 1].max() - t[:, 0].min()) * 0.01):.2f} us")
    for i in range(1, n):
        print(f"   stamp {i}: median {med[i]:6.2f} us   (+{med[i] - med[i - 1]:5.2f})   max {rel[:, i].max():6.2f}")
