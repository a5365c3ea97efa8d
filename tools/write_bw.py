"""Write-only and copy bandwidth of the box (torch fill / copy kernels): the yardstick for the GEMM epilogues' output streams."""
import torch
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from _timing import time_us
for mb in (77, 232, 310, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    y = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    t = time_us(lambda: x.fill_(1.0), 50, 100)
    c = time_us(lambda: y.copy_(x), 50, 100)
    print(f"{mb:5d} MB: fill {t:7.1f} us = {mb*1.048576/t*1e3:7.1f} GB/s written | copy {c:7.1f} us = {2*mb*1.048576/c*1e3:7.1f} GB/s read+written", flush=True)
