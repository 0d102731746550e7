#!/usr/bin/env python
"""Batch-assembly benchmark: device-resident mapped-signal store vs the host path.

    python tools/chunkbench.py [--reads 2000] [--batch 128] [--chunk-len 4000] [--reps 50]
Prints the time of one `MappedSignalStore.sample_chunks` call (candidates drawn on the device,
three kernel launches, no host sync).  The host-path comparison (the numpy restatement of the
reference's per-chunk Python code) lives with the test infrastructure:
`python -m tests.helpers.cpu_legs chunks`.  Algorithmic bytes: 6 per sample (int16 in, float32
out).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taiyaki_amd import mapped_signal as ms, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--chunk-len", type=int, default=4000)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true", help="(kept for old command lines; no effect)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    reads = synth.mapped_reads(args.reads, 7, mean_reflen=900, long_dwell_prob=0.0003)
    t0 = time.time()
    store = ms.MappedSignalStore(reads, dev)
    torch.cuda.synchronize()
    print("store: %d reads, %.1f MB on the device, packed in %.2f s" % (store.nreads, store.nbytes / 1e6,
                                                                        time.time() - t0))
    torch.manual_seed(1)
    fp = store.sample_filter_parameters(1000, args.chunk_len, 3.0, 10.0, 0.5, 5, 1.1)
    N, T = args.batch, args.chunk_len
    for _ in range(3):
        b = store.sample_chunks(N, T, fp)
    assert b.naccepted == N
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.reps):
        b = store.sample_chunks(N, T, fp)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.reps
    print("device  batch of %d x %d: %8.1f us per batch (wall, incl. allocation and draws) = %.0f chunks/s"
          % (N, T, dt * 1e6, N / dt))
    # kernels only: fixed candidates, events around the three C-ABI calls
    cr, _, cf = store._candidates(int(N / 0.5), T, None, None, True, 0)
    evs = []
    for _ in range(args.reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        loc = store._locate(cr, None, cf, T, fp)
        store._select(loc, N)
        e.record()
        evs.append((a, e))
    torch.cuda.synchronize()
    print("        locate + select kernels: %.1f us" % (np.mean([a.elapsed_time(e) for a, e in evs[5:]]) * 1e3))


if __name__ == "__main__":
    main()
