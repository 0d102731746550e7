#!/usr/bin/env python
"""Hash beam search timing (run it under rocprofv3 --kernel-trace --stats for the kernel's own
duration).  Prints per shape the wall time of `decodeutil.beamsearch` (kernel + result copies to
the host) and blocks per second.  (The reference C on one host core, timed through
oracle/_ref in tests/test_beamsearch.py's helpers: 4.36 ms for one read of 2000 blocks = 0.46
Mblocks/s.)

    python tools/beambench.py [--shapes 2000x512,4000x1024,800x128] [--width 5]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taiyaki_amd import decodeutil, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="2000x512,4000x1024,800x128")
    ap.add_argument("--width", type=int, default=5)
    ap.add_argument("--cut", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    for item in args.shapes.split(","):
        T, N = (int(v) for v in item.split("x"))
        sc_h = (synth.scores(T, N, 40, 5) * np.float32(0.8)).astype(np.float32)
        sc = torch.from_numpy(sc_h).cuda()
        decodeutil.beamsearch(sc, args.cut, args.width, True)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.reps):
            decodeutil.beamsearch(sc, args.cut, args.width, True)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.reps
        line = "beam T=%d N=%d width=%d: %.2f ms/call, %.2f Mblocks/s" % (T, N, args.width, dt * 1e3, T * N / dt / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
