#!/usr/bin/env python
"""How much does kernel A's sweep slow down when the posterior pass of OTHER rows runs on the
same CUs?  (Feasibility of fusing the posterior pass into the sweep launch.)  Two streams on two
hardware queues: stream B loops posterior-only launches on a finished lattice, stream A times
sweep-only launches with HIP events.

    GPU_MAX_HW_QUEUES=4 python tools/overlap_probe.py    # (bench.py's import pins it to 1: one queue serialises the two streams)
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taiyaki_amd import _lib  # noqa: E402

_lib.use_lab(True)              # tk_lab_crf_band_phase is a lab-build export


def main():
    dev = torch.device("cuda:0")
    _lib.set_strict(False)
    L = _lib.lib()
    phase = L.tk_lab_crf_band_phase
    phase.argtypes = [ctypes.c_int]
    phase.restype = None
    a = bench.LossOps(800, 128, dev, realistic_chunk_len=4000)
    b = bench.LossOps(800, 128, dev, realistic_chunk_len=4000)
    for ops in (a, b):
        ops.crf()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn, stream, reps):
        evs = []
        with torch.cuda.stream(stream):
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                evs.append((e0, e1))
        return evs

    def report(label, evs):
        us = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
        print("%-46s mean %7.1f us  median %7.1f  min %7.1f" % (label, float(np.mean(us)), us[len(us) // 2], us[0]))

    for ph, name in ((1, "sweep only"), (2, "posterior only"), (0, "both passes")):
        phase(ph)
        ev = timed(a.crf, sa, 30)
        torch.cuda.synchronize()
        report("alone: " + name, ev[5:])
    # contention: B keeps the posterior pass running while A's sweeps are timed
    phase(2)
    with torch.cuda.stream(sb):
        for _ in range(400):
            b.crf()
    phase(1)
    ev = timed(a.crf, sa, 40)
    torch.cuda.synchronize()
    report("sweep while posterior passes run beside it", ev[5:])
    phase(1)
    with torch.cuda.stream(sb):
        for _ in range(200):
            b.crf()
    phase(2)
    ev = timed(a.crf, sa, 60)
    torch.cuda.synchronize()
    report("posterior pass while sweeps run beside it", ev[5:])
    phase(0)


if __name__ == "__main__":
    main()
