#!/usr/bin/env python
"""logZ forward-backward op (C ABI, buffers allocated once) over batch sizes at T = 4000:
roofline fraction = 3 T N S 4 bytes / duration / 8 TB/s.

    python tools/logz_sweep.py [--N 256,320,384,448,512] [--T 4000] [--reps 30]
TK_LOGZ_SPLIT=0 / 1 switches the two-queue pipeline over read halves off / on."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", default="256,320,384,448,512")
    ap.add_argument("--T", type=int, default=4000)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from taiyaki_amd import _lib
    _lib.use_lab(True)          # TK_LOGZ_SPLIT / TK_LOGZ_CH are lab switches
    _lib.set_strict(False)
    for N in [int(x) for x in args.N.split(",")]:
        ops = bench.LossOps(args.T, N, dev)
        mean_s, min_s = bench._events_mean_min(ops.logz_op, args.reps, warm=10)
        alg = 3.0 * args.T * N * 40 * 4
        print("T=%d N=%4d  mean %7.1f us  min %7.1f us   %5.2f TB/s = %.3f of 8 TB/s (min: %.3f)   split=%s"
              % (args.T, N, mean_s * 1e6, min_s * 1e6, alg / mean_s / 1e12, alg / mean_s / 8e12, alg / min_s / 8e12,
                 os.environ.get("TK_LOGZ_SPLIT", "auto")), flush=True)
        del ops
    _lib.raise_if_nonfinite()


if __name__ == "__main__":
    main()
