"""Timing helper of the micro-benchmarks: a kernel is timed over a window of tens of milliseconds after a warm-up of the same
length.  The clocks of an MI355X settle over milliseconds: a 20-launch window (a few ms) measures the state the previous
kernel left behind - the same GEMM read 239 or 320 us depending on what ran before it (round 2) - and such numbers neither
agree with each other nor with the per-kernel times inside a model step."""
import torch


def time_us(fn, warmup: int = 100, iters: int = 150) -> float:
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
