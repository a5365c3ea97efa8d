"""Where a step of the persistent decode kernel spends its time: PM_MI355X_LIB=.../libpm_mi355x_stamps.so python tools/ps_stamps.py
(diagnostic build: `make -C pytorch-models_amd/csrc stamps`).  Prints, for 4 stamped workgroups, per stage of layer 3 the
microseconds between the stamp points, and the stage-to-stage totals averaged over layers."""
import ctypes, os, sys
sys.path[:0] = ["/root/repo", "/root/repo/pytorch-models_amd"]
import numpy as np
import torch
from synthweights import fill_module, synth_input, synth_tokens
from pytorch_models._hip import lib
from pytorch_models.audio2text import Whisper
from pytorch_models.audio2text.generate import GreedyDecoder
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w = Whisper.from_openai("base").eval(); fill_module(w, 56); w = w.to(torch.bfloat16).cuda()
mem = synth_input("m", (B, 1500, 512), 1).to(torch.bfloat16).cuda()
prompt = synth_tokens("p", (B, 4), 51865, 2)
ps = GreedyDecoder(w.decoder, mem, prompt.cuda(), 224, path="persistent")
ps.reset()
for i in range(100):
    ps.step()
torch.cuda.synchronize()
L = lib()
L.pm_dec_layers_stamps.argtypes = [ctypes.c_void_p]
buf = np.zeros((4, 32 * 6, 8), dtype=np.uint64)
assert L.pm_dec_layers_stamps(buf.ctypes.data) == 0
t = buf.astype(np.float64) / 100.0  # us
nl = 8
names = ["self", "so", "cross", "co", "fc1", "fc2"]
for slot in range(4):
    print(f"--- workgroup {slot * 85}")
    base = t[slot, 0, 0]
    for l in (0, 3, 7):
        for s_ in range(6):
            r = t[slot, l * 6 + s_]
            if r[0] == 0:
                print(f" L{l} {names[s_]:5s}: no task"); continue
            d = np.diff(r)
            print(f" L{l} {names[s_]:5s}: enter {r[0]-base:8.2f}  " + " ".join(f"{x:6.2f}" for x in d) + f"   total {r[7]-r[0]:6.2f}")
    ends = [t[slot, l * 6 + 5, 7] for l in range(nl) if t[slot, l * 6 + 5, 7] > 0]
    starts = [t[slot, l * 6, 0] for l in range(nl)]
    print(" layer starts (us):", " ".join(f"{x-base:7.1f}" for x in starts))
