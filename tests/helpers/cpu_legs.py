#!/usr/bin/env python
"""Host-side timing legs for the widened rows (test infrastructure: this is the only place
besides tests/, smoke() and bench.py's cpu_baseline that runs the oracle).  Times the numpy
restatements of the reference's Python code on one core, on the workloads of
tools/chunkbench.py and tools/remapbench.py.

    python -m tests.helpers.cpu_legs chunks [--reads 2000] [--batch 128] [--chunk-len 4000]
    python -m tests.helpers.cpu_legs remap  [--blocks 20000] [--bases 9000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from taiyaki_amd import synth  # noqa: E402


def chunks(args):
    from oracle import chunks as oc
    reads = synth.mapped_reads(args.reads, 7, mean_reflen=900, long_dwell_prob=0.0003)
    rng = np.random.RandomState(3)
    fp = dict(filter_mean_dwell=3.0, filter_max_dwell=10.0, filter_min_pass_fraction=0.5,
              median_meandwell=None, mad_meandwell=None, model_stride=5, path_buffer=1.1)
    med, mad = oc.sample_filter_parameters(reads, 1000, args.chunk_len, fp, rng)
    fp.update(median_meandwell=med, mad_meandwell=mad)
    N, T = args.batch, args.chunk_len
    t0 = time.time()
    reps = 3
    for _ in range(reps):
        cands = oc.candidates_from_rng(reads, int(N / 0.5), T, rng)
        got, _, _ = oc.sample_chunks(reads, N, T, fp, cands)
        oc.assemble_batch(got, 4)
    ct = (time.time() - t0) / reps
    print("host (numpy restatement of the reference's per-chunk path, 1 core): %.1f us per batch of %d x %d"
          " = %.0f chunks/s" % (ct * 1e6, N, T, N / ct))


def remap(args):
    from oracle import remap as orm
    sc = synth.scores(args.blocks, 1, 40, 5)[:, 0, :]
    bases = synth.randint(5, 21, args.bases, 4)
    t0 = time.time()
    score, _ = orm.flipflop_remap(sc, bases, 4, localpen=3.0)
    print("host (numpy restatement of the reference, 1 core): %.1f ms per read of %d blocks x %d bases  score %.3f"
          % ((time.time() - t0) * 1e3, args.blocks, args.bases, score))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["chunks", "remap"])
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--chunk-len", type=int, default=4000)
    ap.add_argument("--blocks", type=int, default=20000)
    ap.add_argument("--bases", type=int, default=9000)
    a = ap.parse_args()
    {"chunks": chunks, "remap": remap}[a.what](a)
