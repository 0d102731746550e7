#!/usr/bin/env python
"""Loss-kernel microbenchmark (run it under rocprofv3 --kernel-trace --stats).

    python tools/microbench.py [--reps 10] [--shapes rowK,cfg2,cfg5] [--ops logz,crf,viterbi]
Prints one line per (op, shape): HIP-event mean / min duration and the
algorithmic-bytes GB/s (3*T*N*S*4 for forward-backward+grad ops,
1*T*N*S*4 for forward-only ops).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taiyaki_amd import _lib, ctc, decode, layers, qscores, synth  # noqa: E402

SHAPES = {"rowK": (4000, 256), "cfg2": (800, 128), "cfg5": (1600, 64), "big": (4000, 1024), "big2k": (4000, 2048),
          "big4k": (4000, 4096), "bigshort": (800, 8192)}


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e6 * max(1, reps // 10)))     # the host enqueues ahead of a busy GPU
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in evs]
    return float(np.mean(ms)) * 1e-3, float(np.min(ms)) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--shapes", default="rowK,cfg2")
    ap.add_argument("--ops", default="logz,logz_fwd,crf,crf_fwd,viterbi")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    _lib.set_strict(False)      # no per-op status sync inside the timed region
    for sh in args.shapes.split(","):
        T, N = SHAPES[sh]
        inp = synth.crf_case(T, N, 1)
        x = torch.from_numpy(inp["scores"]).to(dev)
        seqs, seqlens = torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"])
        nbytes = T * N * 40 * 4
        ops = {
            "logz": (lambda: layers._logz_launch(x, True), 3),
            "logz_fwd": (lambda: layers._logz_launch(x, False), 1),
            "crf": (lambda: ctc._run(x, seqs, seqlens, 1.0, 1.0, 1.0, 40, True), 3),
            "crf_fwd": (lambda: ctc._run(x, seqs, seqlens, 1.0, 1.0, 1.0, 40, False), 1),
            "viterbi": (lambda: decode.flipflop_viterbi(x), 1),
            "viterbi_path": (lambda: decode.flipflop_viterbi_path(x), 1),
        }
        if "catmod" in args.ops.split(","):
            cm = synth.crf_case(T, N, 2, nmods_per_base=(1, 1, 0, 0))
            xm = torch.from_numpy(cm["scores"]).to(dev)
            ms, ml, mc = (torch.from_numpy(cm[k]) for k in ("seqs", "seqlens", "mod_cats"))
            ops["catmod"] = (lambda: ctc._run(xm, ms, ml, 1.0, 1.0, 1.0, 40, True, mc,
                                              cm["can_mods_offsets"], cm["mod_cat_weights"]), 3)
        if "errprobs" in args.ops.split(","):
            trans = decode.flipflop_make_trans(x)
            path = decode.flipflop_viterbi(x)[2]
            ops["errprobs"] = (lambda: qscores.errprobs_from_trans(trans, path), 1)
        for name in args.ops.split(","):
            fn, mult = ops[name]
            mean, mn = timed(fn, args.reps)
            print("%-9s %-5s T=%d N=%d  mean %9.1f us  min %9.1f us  alg %7.1f GB/s (%.1f%% of 8 TB/s)"
                  % (name, sh, T, N, mean * 1e6, mn * 1e6, mult * nbytes / mean / 1e9,
                     mult * nbytes / mean / 8e12 * 100), flush=True)


if __name__ == "__main__":
    main()
    _lib.raise_if_nonfinite()
