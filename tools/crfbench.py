#!/usr/bin/env python
"""Kernel A (sequence CRF) timing by mode and shape (run it under rocprofv3 --kernel-trace
--stats for the per-kernel split).

    python tools/crfbench.py [--reps 10] [--shapes cfg2,cfg2r,cfg5,rowK] [--modes band1,band2,ckpt]
Shapes: cfg2 = T 800 x N 128 with the reference's SPEED_TEST lengths (0.45-0.55 T);
cfg2r = the same with realistic chunk lengths (what the train step launches); cfg5 = T 1600 x
N 64; rowK = T 4000 x N 256.  Algorithmic bytes = 3 T N S 4."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taiyaki_amd import _lib, ctc, synth  # noqa: E402

_lib.use_lab(True)              # the MODES below are lab switches of the dispatch

SHAPES = {"cfg2": (800, 128, None), "cfg2r": (800, 128, 4000), "cfg5": (1600, 64, None),
          "cfg5r": (1600, 64, 8000), "rowK": (4000, 256, None), "rowK8": (4000, 256, "short"), "cfg4": (800, 128, None), "cfg4r": (800, 128, 4000),
          "one": (800, 1, 4000), "short": (800, 128, 450), "short1": (800, 1, 450), "mid": (800, 128, 1100)}
MODES = {"band1": dict(TK_CRF_MODE="band", TK_CRF_BAND_R="1"), "band2": dict(TK_CRF_MODE="band", TK_CRF_BAND_R="2"),
         "band4": dict(TK_CRF_MODE="band", TK_CRF_BAND_R="4"), "band": dict(TK_CRF_MODE="band"),
         "bandnf": dict(TK_CRF_MODE="band", TK_CRF_NO_FALLBACK="1"), "ckpt": dict(TK_CRF_MODE="ckpt"),
         # round 3's arithmetic: 8-step blocks, unbiased weights (the plain CRF ships 12-step blocks with a bias of 3)
         "band8": dict(TK_CRF_MODE="band", TK_CRF_BK="8", TK_CRF_WBIAS="0"),
         "bk4": dict(TK_CRF_MODE="band", TK_CRF_BK="4", TK_CRF_WBIAS="0")}


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(2.0e6 * max(1, reps // 10)))
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
    ap.add_argument("--shapes", default="cfg2,cfg2r,rowK")
    ap.add_argument("--modes", default="band,ckpt")
    ap.add_argument("--fwd", action="store_true", help="also time the cost-only call")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    _lib.set_strict(False)
    for sh in args.shapes.split(","):
        T, N, chunk_len = SHAPES[sh]
        if chunk_len == "short":
            # SPEED_TEST lengths scaled to at most 2048 cells: eight 256-cell chunks at four cells per lane
            seqlens = np.minimum((synth.speedtest_seqlens(T, N) * 0.93).astype(np.int32), 2048)
        else:
            seqlens = None if chunk_len is None else synth.realistic_seqlens(T, N, 17000, chunk_len, 9.0)
        mods = (1, 1, 0, 0) if sh.startswith("cfg4") else None
        inp = synth.crf_case(T, N, 1, seqlens=seqlens, nmods_per_base=mods)
        if mods is not None:
            synth.normalise_mod_columns(inp)
        S = inp["scores"].shape[2]
        x = torch.from_numpy(inp["scores"]).to(dev)
        seqs, seqlens_t = torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"])
        extra = ()
        if mods is not None:
            extra = (torch.from_numpy(inp["mod_cats"]), inp["can_mods_offsets"], inp["mod_cat_weights"])
        ref = None
        for mode in args.modes.split(","):
            for k in ("TK_CRF_MODE", "TK_CRF_BAND_R", "TK_CRF_NO_FALLBACK", "TK_CRF_BK", "TK_CRF_WBIAS"):
                os.environ.pop(k, None)
            os.environ.update(MODES[mode])
            if mode in ("band1", "band2", "band4"):
                R = int(mode[4:])
                if int(inp["seqlens"].max()) > 1024 * R:
                    continue
            for want_grad in ((True, False) if args.fwd else (True,)):
                fn = lambda: ctc._run(x, seqs, seqlens_t, 1.0, 1.0, 1.0, 40, want_grad, *extra)  # noqa: E731
                cost, grad = fn()
                torch.cuda.synchronize()
                note = ""
                if want_grad:
                    if ref is None:
                        ref = (cost.clone(), grad.clone())
                    else:
                        note = "  dcost %.2e dgrad %.2e vs %s" % (
                            float((cost - ref[0]).abs().max()), float((grad - ref[1]).abs().max()),
                            args.modes.split(",")[0])
                mean, mn = timed(fn, args.reps)
                alg = (3 if want_grad else 1) * T * N * S * 4
                print("%-8s %-6s %s T=%d N=%d maxL=%d  mean %9.1f us  min %9.1f us  alg %7.1f GB/s (%.2f%% of 8 TB/s)%s"
                      % (mode, sh, "grad" if want_grad else "cost", T, N, int(inp["seqlens"].max()), mean * 1e6,
                         mn * 1e6, alg / mean / 1e9, alg / mean / 8e12 * 100, note), flush=True)
    _lib.raise_if_nonfinite()


if __name__ == "__main__":
    main()
