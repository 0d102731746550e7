#!/usr/bin/env python
"""flipflop_remap benchmark.

    python tools/remapbench.py [--blocks 20000] [--bases 9000] [--reads 256] [--cpu-reads 1]
One read alone (latency of the serial time loop + traceback) and a batch of reads in one launch
(one workgroup per read), against the numpy restatement of the reference (oracle/remap.py: the
same ~12 numpy calls per time step as taiyaki/flipflop_remap.py) on one host core.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taiyaki_amd import flipflop_remap as fr, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=20000)
    ap.add_argument("--bases", type=int, default=9000)
    ap.add_argument("--reads", type=int, default=256)
    ap.add_argument("--cpu-reads", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    T, M = args.blocks, args.bases
    sc = torch.tensor(synth.scores(T, 1, 40, 5)[:, 0, :], device=dev)
    bases = synth.randint(5, 21, M, 4)
    step, stay = fr.remap_indices(bases)
    fr.map_to_crf_viterbi(sc, step, stay, 3.0)
    torch.cuda.synchronize()
    t0 = time.time()
    reps = 5
    for _ in range(reps):
        score, path = fr.map_to_crf_viterbi(sc, step, stay, 3.0)
    one = (time.time() - t0) / reps
    print("one read   T=%d M=%d: %8.2f ms (incl. index upload, path download)  score %.3f" % (T, M, one * 1e3, score))
    n = args.reads
    scs = [sc] * n
    fr.map_to_crf_viterbi_batch(scs[:2], [step] * 2, [stay] * 2, 3.0)
    torch.cuda.synchronize()
    t0 = time.time()
    s, p = fr.map_to_crf_viterbi_batch(scs, [step] * n, [stay] * n, 3.0)
    bt = time.time() - t0
    assert np.all(s == score) and all(np.array_equal(x, path) for x in p[:4])
    print("batch of %d: %8.2f ms = %.2f ms per read = %.0f reads/s (%.1f M blocks/s)"
          % (n, bt * 1e3, bt / n * 1e3, n / bt, n * T / bt / 1e6))
    if args.cpu_reads:
        from oracle import remap as orm
        h = sc.cpu().numpy()
        t0 = time.time()
        for _ in range(args.cpu_reads):
            ws, wp = orm.map_to_crf_viterbi(h, step, stay, 3.0)
        ct = (time.time() - t0) / args.cpu_reads
        assert ws == score and np.array_equal(wp, path)
        print("host (numpy restatement of the reference, 1 core): %8.1f ms per read -> GPU %.0fx alone, %.0fx batched"
              % (ct * 1e3, ct / one, ct / (bt / n)))


if __name__ == "__main__":
    main()
