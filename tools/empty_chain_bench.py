"""What does one more launch cost inside a replayed HIP graph?  A chain of N launches of a (nearly) empty kernel, graph-replayed:
the per-launch floor the decode step is measured against (1.57 us).     python tools/empty_chain_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-models_amd")]
import torch
from pytorch_models._hip import lib
pos = torch.zeros(1, dtype=torch.int32, device="cuda")
L = lib()
st = torch.cuda.current_stream().cuda_stream
L.pm_dec_advance(pos.data_ptr(), st)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s2 = torch.cuda.current_stream().cuda_stream
    for i in range(200):
        L.pm_dec_advance(pos.data_ptr(), s2)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
print("empty-ish kernel in a graph chain: %.2f us per launch" % (e0.elapsed_time(e1) / 20 / 200 * 1e3))
