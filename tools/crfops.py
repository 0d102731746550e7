#!/usr/bin/env python
"""Kernel A and the fused loss at the train step's shape (and optionally T=4000/N=256), C entry points back to
back on the stream -- bench.py's `roofline_crf` / `loss_path` timing without the train step around it.  For lab
switches (TK_CRF_BK, TK_CRF_WBIAS, TK_CRF_FEED, TK_CRF_BAND_R ...) and lab builds (TAIYAKI_AMD_LIB).

    python tools/crfops.py [--rowk] [--catmod] [--cfg5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402
from taiyaki_amd import _lib  # noqa: E402

if any(k.startswith(("TK_CRF_", "TK_LOGZ_", "TK_K1_")) for k in os.environ):
    _lib.use_lab(True)          # the dispatch switches only exist in the lab build


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rowk", action="store_true")
    ap.add_argument("--catmod", action="store_true")
    ap.add_argument("--cfg5", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--separate", action="store_true", help="round 4's form: tk_flipflop_build_indices_dev as a launch of its own")
    ap.add_argument("--shapes", default="", help="extra shapes name:T:N:chunk_len[:spb[:cm]], comma separated (cm: the cat-mod form; (chunk_len 0 = SPEED_TEST lengths)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    _lib.lib()
    _lib.set_strict(False)
    shapes = [("step", 800, 128, 4000, 9.0, False)]
    if args.catmod:
        shapes.append(("catmod", 800, 128, 4000, 9.0, True))
    if args.cfg5:
        shapes.append(("cfg5", 1600, 64, 8000, 9.5, False))
    if args.rowk:
        shapes.append(("rowK", 4000, 256, None, 9.0, False))
    for item in [x for x in args.shapes.split(",") if x]:
        f = item.split(":")
        shapes.append((f[0], int(f[1]), int(f[2]), int(f[3]) or None, float(f[4]) if len(f) > 4 and f[4] else 9.0,
                       len(f) > 5 and f[5] == "cm"))
    out = []
    for name, T, N, cl, spb, cm in shapes:
        ops = bench.LossOps(T, N, dev, realistic_chunk_len=cl, spb=spb, cat_mod=cm)
        ops.separate_index_build = True      # (once: the index arrays exist whatever a lab variant of the in-launch build leaves out)
        ops.crf()
        torch.cuda.synchronize()
        ops.separate_index_build = args.separate
        reps = args.reps if T < 4000 else 5
        crf, crf_min = bench._events_mean_min(ops.crf, reps, warm=5)
        both, _ = bench._events_mean_min(ops.both, reps, warm=5)
        assert ops.finite()
        out.append("%s[L<=%d]: crf %.1f us (min %.1f)  fused loss %.1f us" % (name, ops.maxlen, crf * 1e6, crf_min * 1e6, both * 1e6))
    print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
