#!/usr/bin/env python
"""logZ op (three launches) timed as plain launches vs replayed from a hipGraph (what the train step
does with it): HIP events around each."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taiyaki_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    _lib.set_strict(False)
    for T, N in ((4000, 256), (800, 128)):
        ops = bench.LossOps(T, N, dev)
        for name, fn in (("logz", ops.logz_op), ("crf", ops.crf), ("both", ops.both)):
            m0, n0 = bench._events_mean_min(fn, 40)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            torch.cuda.synchronize()
            m1, n1 = bench._events_mean_min(g.replay, 40)
            print("T=%d N=%d %-5s plain %.1f us (min %.1f)   graph replay %.1f us (min %.1f)"
                  % (T, N, name, m0 * 1e6, n0 * 1e6, m1 * 1e6, n1 * 1e6))


if __name__ == "__main__":
    main()
