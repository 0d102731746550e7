#!/usr/bin/env python
"""Random sweep of the hash beam search kernel against the oracle (GPU box; test infrastructure).

    python -m tests.helpers.fuzz_beam [--cases 60] [--seed 1]

Every case draws T, beam width, beam cut, guided / unguided and a score distribution: continuous
(ties at the edge of the beam are rare), on a grid of 0.5 or 0.25 (ties in nearly every block: the
kernel's restatement of the reference's sort procedure decides), or saturated (5 tanh of wide
normals: long runs of equal scores).  The oracle runs with the kernel's rounding of exp / log1p
(oracle.beam.MATH = "cr"); sequences and float scores must agree bit for bit.  Batches of a few
reads per launch, so reads of one launch must not see each other.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import beam  # noqa: E402
from taiyaki_amd import decodeutil  # noqa: E402


def case_scores(rng, T, N, kind):
    x = rng.randn(T, N, 40).astype(np.float32) * np.float32(rng.uniform(0.5, 3.0))
    if kind == "grid":
        step = np.float32(rng.choice([0.5, 0.25]))
        x = (np.round(x / step) * step).astype(np.float32)
    elif kind == "saturated":
        x = (5 * np.tanh(x * 3)).astype(np.float32)
    return x


def case(k, rng, dev):
    T = int(rng.choice([1, 2, 3, 7, 33, 64, 100, 180, 300]))
    N = int(rng.randint(1, 5))
    w = int(rng.choice([1, 2, 3, 5, 5, 8, 12]))
    cut = float(rng.choice([0.0, 0.0, 1e-4, 0.02, 0.5]))
    guided = bool(rng.randint(2))
    kind = str(rng.choice(["continuous", "grid", "saturated"]))
    sc = case_scores(rng, T, N, kind)
    seqs, scores = decodeutil.beamsearch(torch.from_numpy(sc).to(dev), cut, w, guided)
    ok = True
    for n in range(N):
        ws, wsc = beam.beamsearch(np.ascontiguousarray(sc[:, n]), cut, w, guided)
        ok = ok and np.array_equal(seqs[n], ws) and np.float32(scores[n]) == np.float32(wsc)
    return ok, "T=%3d N=%d width=%2d cut=%-6g %s %s" % (T, N, w, cut, "guided  " if guided else "unguided", kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    beam.MATH = "cr"
    rng = np.random.RandomState(args.seed)
    dev = torch.device("cuda:0")
    bad = 0
    for k in range(args.cases):
        ok, msg = case(k, rng, dev)
        bad += not ok
        print("%s %s" % ("ok  " if ok else "FAIL", msg), flush=True)
    print("fuzz_beam: %d cases, %d failures" % (args.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
