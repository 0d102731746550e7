#!/usr/bin/env python
"""Random-shape sweep of the three lattice operators against the oracle (GPU box; test
infrastructure).  Shapes straddle the launcher's regime switches: cooperative / one-wave / LDS-ring
transfer kernels (ncols x chunks around 640 and 900), 8- vs 16-row chunks (around 384), partial
columns (N % 64), ragged last chunks, single rows.

    python -m tests.helpers.fuzz_shapes [--cases 40] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from taiyaki_amd import synth  # noqa: E402
from tests import parity  # noqa: E402


EDGES = [(1, 1), (15, 64), (16, 65), (17, 63), (639 * 16 // 10, 640), (2048, 300), (2049, 300),
         (1536, 384), (1535, 385), (2400, 320), (3601, 256), (100, 1000), (9, 2048)]


def loss_ok(c):
    """Every read: 1e-4 relative (north_star; the fp32 reference itself carries ~1e-5 at T ~ 2000 for
    one-base sequences) OR 2e-6 absolute -- a read whose path scores cancel to a loss of ~1e-5 (T = 2237,
    L = 1: kernel 2.3772e-05, fp32 oracle 2.3796e-05, float64 2.3768e-05, tests/helpers/fuzz_debug.py)
    has no relative accuracy to speak of in fp32.  Per read: the largest relative and the largest
    absolute error of a batch usually sit in different reads."""
    loss, oloss = np.asarray(c["loss"], dtype=np.float64), np.asarray(c["oloss"], dtype=np.float64)
    err = np.abs(loss - oloss)
    return bool(np.all((err <= 1e-4 * np.abs(oloss)) | (err < 2e-6)))


def case(k, rng, dev, oracle_mod=None):
    """Case k of the sweep (the first len(EDGES) are the regime-switch shapes, the rest random).
    Returns (ok, one-line description)."""
    oracle_mod = oracle_mod or oracle
    if k < len(EDGES):
        T, N = EDGES[k]
    else:
        T, N = int(rng.randint(1, 2600)), int(rng.randint(1, 420))
    sc = synth.scores(T, N, 40, 7000 + k)
    r = parity.compare_logz(oracle_mod, sc, dev)
    v = parity.compare_viterbi(oracle_mod, sc, dev)
    L = max(1, min(T - 1, int(T * rng.uniform(0.05, 0.85)))) if T > 1 else 1
    # a zero-length read is only put LAST: the reference's move-index layout gives every read
    # L - 1 slots (c_crf_flipflop.c:479-480), i.e. minus one for an empty read, so an empty read
    # in the middle makes its neighbours' slots overlap -- in the reference and in its oracle
    seqlens = np.clip(rng.randint(1, L + 1, size=N), 1, max(T, 1)).astype(np.int32)
    if k % 3 == 0:
        seqlens[-1] = 0
    inp = synth.crf_case(T, N, 7100 + k, seqlens=seqlens)
    inp["scores"] = sc
    c = parity.compare_crf(oracle_mod, inp, 1.0, dev)
    # the cat-mod variant of the same kernels on every third case (46 columns, own scores)
    cm_ok, cm = True, None
    if k % 3 == 1 and T > 1:
        minp = synth.crf_case(T, N, 7200 + k, nmods_per_base=(1, 1, 0, 0), seqlens=seqlens)
        cm = parity.compare_crf(oracle_mod, minp, 1.0, dev)
        cm_ok = cm["finite"] and loss_ok(cm) and cm["grad_abs"] < 5e-5 and parity.crf_grad_ok(cm)
    # trained-network-like scores (one alignment per read at +4, everything else at -3; steady or
    # bursty strands) on every second case, sharpened a little on every fourth: the posterior mass
    # sits on one path, which is where the linear band path's frames, skips and row totals are tried
    cf_ok, cf = True, None
    if k % 2 == 1 and T > 1:
        finp = synth.crf_case(T, N, 7300 + k, seqlens=seqlens)
        synth.confident_scores(finp, 7400 + k, bursty=(k % 4 == 3))
        cf = parity.compare_crf(oracle_mod, finp, 1.25 if k % 8 == 5 else 1.0, dev)
        cf_ok = cf["finite"] and loss_ok(cf) and cf["grad_abs"] < 2e-5 and parity.crf_grad_ok(cf)
    ok = cf_ok and cm_ok and (r["finite"] and r["logz_rel"] < 1e-5 and r["grad_abs"] < 2e-5 and r["nograd_same"] == 0.0 and
                    v["path_mismatch"] == 0 and v["fwd_bit_mismatch"] == 0 and
                    c["finite"] and loss_ok(c) and c["grad_abs"] < 2e-5 and parity.crf_grad_ok(c))
    # (CRF gradient errors on the posterior scale: kernel vs float64 witness [fp32 oracle vs witness])
    g = lambda x: "%.1e [%.1e]" % (x["grad_f64_scaled"], x["ref_noise_scaled"])     # noqa: E731
    msg = "T=%4d N=%4d  logz %.1e / %.1e  viterbi %d  crf %.1e (abs %.1e) / %s" % (
        T, N, r["logz_rel"], r["grad_abs"], v["path_mismatch"], c["loss_rel"], c["loss_abs"], g(c)) + (
        "  catmod %.1e / %s" % (cm["loss_rel"], g(cm)) if cm else "") + (
        "  confident %.1e / %s" % (cf["loss_rel"], g(cf)) if cf else "")
    return bool(ok), msg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    dev = torch.device("cuda:0")
    bad = 0
    for k in range(args.cases):
        ok, msg = case(k, rng, dev)
        bad += not ok
        print("%s %s" % ("ok  " if ok else "FAIL", msg), flush=True)
    print("fuzz: %d cases, %d failures" % (args.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
