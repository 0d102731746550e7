#!/usr/bin/env python
"""Do streams of different priority share a hardware queue when GPU_MAX_HW_QUEUES=1?

    GPU_MAX_HW_QUEUES=1 python tools/queue_probe.py
A long spin kernel goes to a normal-priority stream, then a tiny kernel to (a) another
normal-priority stream and (b) a high-priority stream; the tiny kernel finishing while the spin
kernel still runs means it sits in a different hardware queue.
"""
import os
import time

import torch


def probe(prio):
    torch.cuda.synchronize()
    a = torch.cuda.Stream()
    b = torch.cuda.Stream(priority=prio)
    x = torch.zeros(1024, device="cuda")
    done = torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(int(2.0e9))          # ~1 s of spinning
    time.sleep(0.05)
    with torch.cuda.stream(b):
        x.add_(1.0)
        done.record()
    t0 = time.time()
    while not done.query() and time.time() - t0 < 5.0:
        time.sleep(0.001)
    dt = time.time() - t0
    torch.cuda.synchronize()
    return dt


if __name__ == "__main__":
    torch.zeros(1, device="cuda")
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
    lo, hi = torch.cuda.Stream.priority_range()
    print("priority range", lo, hi)
    print("tiny kernel on another NORMAL stream finished after %.3f s" % probe(0))
    print("tiny kernel on a HIGH-priority stream finished after %.3f s" % probe(hi))
