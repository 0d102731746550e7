#!/usr/bin/env python
"""What streaming bandwidth does this MI355X deliver for plain torch kernels?"""
import torch

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2] * 1e-3, ms[0] * 1e-3


for mb in (164, 655, 2048):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("read  (sum)", lambda: x.sum(), 4 * n),
                             ("write (fill)", lambda: y.fill_(1.0), 4 * n),
                             ("copy", lambda: y.copy_(x), 8 * n),
                             ("mul (r+w)", lambda: torch.mul(x, 2.0, out=y), 8 * n)):
        med, mn = timed(fn)
        print("%5d MB %-13s median %8.1f us  min %8.1f us  -> %6.2f TB/s (min %6.2f)"
              % (mb, name, med * 1e6, mn * 1e6, nbytes / med / 1e12, nbytes / mn / 1e12), flush=True)
