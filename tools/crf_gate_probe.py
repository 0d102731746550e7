#!/usr/bin/env python
"""Which reads does kernel A's linear band path disown (gate -> log-domain crf_kernel), and how far is
it from the checkpoint kernel on the reads it keeps?

    python tools/crf_gate_probe.py [--shapes cfg2,cfg2r,cfg5,rowK,narrow,sharp,realnet,realfast_cm_s2]
Runs every shape three times: TK_CRF_MODE=ckpt (the log-domain form on every read = the reference
arithmetic), band with TK_CRF_NO_FALLBACK=1 (the linear path alone: the batch's launch AND the tail launch's
per-read retry, no log-domain redo) and band as shipped.  "gated" = reads neither linear attempt kept."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taiyaki_amd import _lib, ctc, synth  # noqa: E402

_lib.use_lab(True)              # TK_CRF_MODE / TK_CRF_NO_FALLBACK only exist in the lab build

SHAPES = {
    "tiny": (20, [9, 1, 21, 20, 2, 0], 1.0, None),
    "t37": (37, [12, 30, 38, 5], 1.0, None),
    "t200": (200, [90, 150, 201, 30, 195, 180], 1.0, None),
    "cfg2": (800, "speed128", 1.0, None),
    "cfg2r": (800, "real128", 1.0, None),
    "narrow": (800, [400, 533, 380, 700, 780, 800, 801, 790, 760, 100, 5, 1], 1.0, None),
    # around the reference's path-buffer filter (signal_mapping.py:699-703: a chunk needs T / L > 1.1, L <= 727 at T = 800)
    "pathbuf": (800, [700, 710, 720, 727, 735, 745, 750, 755], 1.0, None),
    "lenramp": (800, [560, 600, 640, 660, 680, 690, 700, 705, 710, 715], 1.0, None),
    "conframp": (800, [560, 600, 640, 660, 680, 700, 710, 720, 727, 740, 760, 780], 1.0, None),
    # a freshly initialised network: 5 tanh of small activations, |score| <= 1
    "initramp": (800, [560, 640, 680, 700, 710, 720, 727, 740, 760, 780, 795, 801], 1.0, None),
    "sharp": (800, "speed32", 2.5, None),
    # the reference's --sharpen schedule (bin/_bin_argparse.py:58-62): 8-step blocks up to 1.36, biased weights up
    # to 1.76, 4-step blocks up to 3.5, the log-domain kernel beyond
    "sharp13": (800, "real32", 1.3, None), "sharp15": (800, "real32", 1.5, None), "sharp17": (800, "real32", 1.75, None),
    "sharp2": (800, "real32", 2.0, None), "sharp3": (800, "real32", 3.0, None), "sharp4": (800, "real32", 4.0, None),
    "sharp2K": (4000, "speed16", 2.0, None), "cfg4sharp2": (800, "real32", 2.0, (1, 1, 0, 0)),
    "cfg4sharp25": (800, "real32", 2.5, (1, 1, 0, 0)), "cfg4sharp3": (800, "real32", 3.0, (1, 1, 0, 0)),
    "cfg4": (800, "speed128", 1.0, (1, 1, 0, 0)),
    "cfg4w1": (800, "speed128", 1.0, (1, 1, 0, 0)),
    "cfg4r": (800, "real128", 1.0, (1, 1, 0, 0)),
    "cfg4rharsh": (800, "real128", 1.0, (1, 1, 0, 0)),     # log-probabilities down to -10 x weight 8, iid per row
    "cfg4free": (800, "speed128", 1.0, (1, 1, 0, 0)),      # free U(-5, 5) mod scores x weight 8: overflows
    "cfg5": (1600, "speed64", 1.0, None),
    "rowK": (4000, "speed256", 1.0, None),
    "t19": (19, [20, 3], 1.0, None),
    # a trained network: one alignment per read scores +4, everything else -3
    "conf": (800, "real128", 1.0, None),
    "confburst": (800, "real128", 1.0, None),
    "confK": (4000, "speed64", 1.0, None),
    # round 6: a TRAINED network's scores on real reads (tests/golden/realnet.npz): "real" L = 0.33 .. 0.50 T,
    # "fast" L = 0.62 .. 0.81 T; x sharpening; x cat-mod (synthetic modification columns beside them)
    "realnet": ("real", 1.0, False), "realnet_s2": ("real", 2.0, False), "realnet_s3": ("real", 3.0, False),
    "realfast": ("fast", 1.0, False), "realfast_s2": ("fast", 2.0, False), "realfast_s3": ("fast", 3.0, False),
    "realnet_cm": ("real", 1.0, True), "realnet_cm_s13": ("real", 1.3, True), "realnet_cm_s2": ("real", 2.0, True),
    "realnet_cm_s25": ("real", 2.5, True), "realnet_cm_s3": ("real", 3.0, True),
    "realfast_cm": ("fast", 1.0, True), "realfast_cm_s13": ("fast", 1.3, True), "realfast_cm_s2": ("fast", 2.0, True),
    "realfast_cm_s25": ("fast", 2.5, True), "realfast_cm_s3": ("fast", 3.0, True),
}
REALNET = [k for k in SHAPES if k.startswith("real")]


def realnet_case(spec):
    from tests.golden import cases
    tag, sharp, catmod = spec
    gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "realnet.npz"))
    if catmod:
        inp = cases.realnet_catmod_inputs(gold, tag)
    else:
        inp = dict(scores=gold[tag + "/scores"], seqs=gold[tag + "/seqs"].astype(np.int64),
                   seqlens=gold[tag + "/seqlens"].astype(np.int32))
    return inp, sharp, (1, 1, 0, 0) if catmod else None


def run(x, seqs, seqlens, sharp, extra, env):
    for k in ("TK_CRF_MODE", "TK_CRF_NO_FALLBACK", "TK_CRF_NO_RETRY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    # poison what the caching allocator will hand out for the outputs: a read nobody computes shows as NaN
    junk = [torch.full_like(x, float("nan")), torch.full((x.shape[1],), float("nan"), device=x.device)]
    del junk
    if extra:
        cost, grad = ctc._run(x, seqs, seqlens, sharp, 1.0, 1.0 / sharp, 40, True, *extra)
    else:
        cost, grad = ctc._run(x, seqs, seqlens, sharp, sharp, 1.0 / sharp, x.shape[2], True)
    torch.cuda.synchronize()
    return cost.cpu().numpy(), grad.cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="tiny,t19,t37,t200,cfg2,cfg2r,narrow,pathbuf,lenramp,initramp,conframp,sharp,sharp13,sharp15,sharp17,sharp2,sharp3,sharp4,sharp2K,cfg4,cfg4r,cfg4rharsh,cfg4sharp2,cfg4sharp25,cfg4sharp3,cfg5,rowK,conf,confburst,confK," + ",".join(REALNET))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    _lib.set_strict(False)
    for sh in args.shapes.split(","):
        if sh in REALNET:
            inp, sharp, mods = realnet_case(SHAPES[sh])
            T, N = inp["scores"].shape[:2]
            lens = []
        else:
            T, lens, sharp, mods = SHAPES[sh]
        if sh in REALNET:
            pass
        elif isinstance(lens, str):
            N = int(lens[5:] if lens.startswith("speed") else lens[4:])
            seqlens = None if lens.startswith("speed") else synth.realistic_seqlens(T, N, 17000, T * 5, 9.0)
        else:
            N, seqlens = len(lens), np.array(lens, dtype=np.int32)
        if sh not in REALNET:
            inp = synth.crf_case(T, N, 1, seqlens=seqlens, nmods_per_base=mods)
        if sh.startswith("init"):
            inp["scores"] *= np.float32(0.2)
        if sh.startswith("conf"):
            synth.confident_scores(inp, 7, bursty="burst" in sh)
        if sh.startswith("sharp") and len(sh) > 5:
            synth.confident_scores(inp, 9, bursty=False)        # a sharpening schedule is applied to a TRAINED network
        if mods is not None and not sh.endswith("free") and sh not in REALNET:
            synth.normalise_mod_columns(inp, logit_scale=1.0 if sh.endswith("harsh") else 0.2)
        x = torch.from_numpy(inp["scores"]).to(dev)
        seqs, sl = torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"])
        extra = ()
        if mods is not None:
            wts = inp["mod_cat_weights"] * (0.125 if sh.endswith("w1") else 1.0)
            extra = (torch.from_numpy(inp["mod_cats"]), inp["can_mods_offsets"], wts)
        c0, g0 = run(x, seqs, sl, sharp, extra, dict(TK_CRF_MODE="ckpt"))
        c1, g1 = run(x, seqs, sl, sharp, extra, dict(TK_CRF_MODE="band", TK_CRF_NO_FALLBACK="1"))
        c2, g2 = run(x, seqs, sl, sharp, extra, dict(TK_CRF_MODE="band"))
        with np.errstate(all="ignore"):
            kept = np.isfinite(c1) & np.array([np.isfinite(g1[:, n]).all() for n in range(N)])
            rel = np.abs(c2 - c0) / np.maximum(np.abs(c0), 1e-30)
            gd = np.array([np.abs(g2[:, n] - g0[:, n]).max() for n in range(N)])
            relk = np.abs(c1 - c0) / np.maximum(np.abs(c0), 1e-30)
            gdk = np.array([np.abs(g1[:, n] - g0[:, n]).max() for n in range(N)])
        print("%-7s T=%d N=%d sharp=%.1f: gated %d / %d reads; shipped vs ckpt: cost rel %.2e grad %.2e; "
              "kept reads, linear path alone: cost rel %.2e grad %.2e"
              % (sh, T, N, sharp, int((~kept).sum()), N, np.nanmax(rel), np.nanmax(gd),
                 np.nanmax(np.where(kept, relk, 0)), np.nanmax(np.where(kept, gdk, 0))), flush=True)
        if (~kept).any() and N <= 16:
            print("        gated reads: lengths", inp["seqlens"][~kept].tolist())
    try:
        _lib.raise_if_nonfinite()
    except Exception as e:                      # noqa: BLE001
        print("status:", type(e).__name__, str(e)[:80])


if __name__ == "__main__":
    main()
