#!/usr/bin/env python
"""flipflop_remap benchmark.

    python tools/remapbench.py [--blocks 20000] [--bases 9000] [--reads 256] 
One read alone (latency of the serial time loop + traceback) and a batch of reads in one launch
(one workgroup per read).  The host-path comparison (numpy restatement of the reference: the
same ~12 numpy calls per time step as taiyaki/flipflop_remap.py) lives with the test
infrastructure: `python -m tests.helpers.cpu_legs remap`.
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
    ap.add_argument("--cpu-reads", type=int, default=0, help="(kept for old command lines; no effect)")
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
    bts = []
    for _ in range(3):                  # (the first full-size call also pays for its workspaces)
        torch.cuda.synchronize()
        t0 = time.time()
        s, p = fr.map_to_crf_viterbi_batch(scs, [step] * n, [stay] * n, 3.0)
        bts.append(time.time() - t0)
    print("batch of %d, call by call: %s ms" % (n, ", ".join("%.1f" % (b * 1e3) for b in bts)))
    bt = min(bts)
    assert np.all(s == score) and all(np.array_equal(x, path) for x in p[:4])
    print("batch of %d: %8.2f ms = %.2f ms per read = %.0f reads/s (%.1f M blocks/s)"
          % (n, bt * 1e3, bt / n * 1e3, n / bt, n * T / bt / 1e6))


if __name__ == "__main__":
    main()
