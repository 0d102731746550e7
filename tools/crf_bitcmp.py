#!/usr/bin/env python
"""Bit-for-bit regression check between two BUILDS of the kernel library (used for every change to kernel A
that must not change arithmetic): costs and gradients of ten shapes -- the train step's, SPEED_TEST
lengths, ragged / degenerate reads, R = 2 and R = 4 launches, cat-mod, a last block of three rows, a batch
larger than the chip, sharpened and trained-network-like scores -- under the library TAIYAKI_AMD_LIB names.

    TAIYAKI_AMD_LIB=/path/to/old.so python tools/crf_bitcmp.py /tmp/old.npz
    python tools/crf_bitcmp.py /tmp/new.npz /tmp/old.npz        # prints which arrays differ (none expected)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taiyaki_amd import ctc, synth, _lib
_lib.set_strict(False)
dev = torch.device("cuda:0")
out = {}
cases = [(800, 128, "real", None, 1.0), (800, 128, "speed", None, 1.0), (200, 7, [90, 150, 201, 30, 195, 64, 65], None, 1.0),
         (1600, 64, "real", None, 1.0), (800, 128, "real", (1, 1, 0, 0), 1.0), (37, 4, [12, 30, 38, 5], None, 1.0),
         (803, 9, "speed", None, 1.0), (4000, 40, "speed", None, 1.0), (800, 300, "real", None, 1.0), (800, 32, "speed", None, 1.2)]
for k, (T, N, lens, mods, sharp) in enumerate(cases):
    seqlens = synth.realistic_seqlens(T, N, 17000, T * 5, 9.0) if lens == "real" else (None if lens == "speed" else np.array(lens, dtype=np.int32))
    inp = synth.crf_case(T, N, 3 + k, seqlens=seqlens, nmods_per_base=mods)
    if k % 3 == 1:
        synth.confident_scores(inp, 5 + k, bursty=True)
    x = torch.from_numpy(inp["scores"]).to(dev)
    seqs, sl = torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"])
    if mods is not None:
        synth.normalise_mod_columns(inp, logit_scale=0.2)
        x = torch.from_numpy(inp["scores"]).to(dev)
        c, g = ctc._run(x, seqs, sl, sharp, 1.0, 1.0 / sharp, 40, True, torch.from_numpy(inp["mod_cats"]), inp["can_mods_offsets"], inp["mod_cat_weights"])
    else:
        c, g = ctc._run(x, seqs, sl, sharp, sharp, 1.0 / sharp, x.shape[2], True)
    torch.cuda.synchronize()
    out["c%d" % k], out["g%d" % k] = c.cpu().numpy(), g.cpu().numpy()
np.savez(sys.argv[1], **out)
if len(sys.argv) > 2:
    ref = np.load(sys.argv[2])
    bad = [k for k in out if not np.array_equal(out[k], ref[k], equal_nan=True)]
    print("bitcmp: %d arrays, different: %s" % (len(out), bad or "none"))
