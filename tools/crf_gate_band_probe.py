#!/usr/bin/env python
"""How narrow a band (L / T) does kernel A's linear path keep?  Reads of L = frac x T (+- 8) bases under iid U(-5, 5) scores, plain CRF and
cat-mod (log-softmax modification columns, random labels): reads disowned (redone by the log-domain kernel) of 64, per L / T.

    python tools/crf_gate_band_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from taiyaki_amd import ctc, synth, _lib
dev = torch.device("cuda:0")
for T in (800, 1600):
    for mods in (None, (1, 1, 0, 0)):
        row = []
        for frac in (0.5, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9):
            N = 64
            rng = np.random.default_rng(7)
            Ls = np.clip((frac * T + rng.integers(-8, 9, N)).astype(np.int32), 1, T)
            inp = synth.crf_case(T, N, 3, seqlens=Ls, nmods_per_base=mods)
            if mods is not None:
                synth.normalise_mod_columns(inp)
            x = torch.from_numpy(inp["scores"]).to(dev)
            seqs, sl = torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"])
            if mods is not None:
                c, g = ctc._run(x, seqs, sl, 1.0, 1.0, 1.0, 40, True, torch.from_numpy(inp["mod_cats"]), inp["can_mods_offsets"], inp["mod_cat_weights"])
            else:
                c, g = ctc._run(x, seqs, sl, 1.0, 1.0, 1.0, 40, True)
            torch.cuda.synchronize()
            row.append("%.2f:%d/%d" % (frac, ctc.last_gate_count(), ctc.last_retry_count()))
        print("T %d %s  of 64, by L/T: redone in the log domain / retried on the linear path  %s" % (T, "cat-mod" if mods else "plain  ", "  ".join(row)), flush=True)
# a batch of ordinary reads with ONE (and with four) long ones: the batch keeps its fast configuration, the long reads are retried
import time
MODES = (("as shipped", {}), ("lab build, TK_CRF_NO_RETRY=1: round 5's path, disowned reads straight to the log domain", {"TK_CRF_NO_RETRY": "1"}))
for T, N, label, env in [(800, 128) + m for m in MODES] + [(2400, 64) + m for m in MODES]:
    print("-- T %d N %d, %s" % (T, N, label), flush=True)
    _lib.use_lab(bool(env))
    for k in ("TK_CRF_NO_RETRY",):
        os.environ.pop(k, None)
    os.environ.update(env)
    for mods in (None, (1, 1, 0, 0)):
        for nlong, frac in ((0, 0.0), (1, 0.86), (1, 0.9), (1, 0.93), (4, 0.93), (8, 0.93)):
            Ls = synth.realistic_seqlens(T, N, 17000, T * 5, 9.0).copy()
            Ls[:nlong] = int(frac * T) - np.arange(nlong)
            inp = synth.crf_case(T, N, 3, seqlens=Ls, nmods_per_base=mods)
            if mods is not None:
                synth.normalise_mod_columns(inp)
            x = torch.from_numpy(inp["scores"]).to(dev)
            seqs, sl = torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"])
            extra = (torch.from_numpy(inp["mod_cats"]), inp["can_mods_offsets"], inp["mod_cat_weights"]) if mods is not None else ()
            def call():
                return ctc._run(x, seqs, sl, 1.0, 1.0, 1.0, 40, True, *extra)
            call(); torch.cuda.synchronize()
            g, r = ctc.last_gate_count(), ctc.last_retry_count()
            _lib.set_strict(False)
            for _ in range(3): call()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): call()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            _lib.take_gate_count(); _lib.set_strict(True)
            print("T %d N %d %s  %d reads of %.2f T among ordinary ones (longest %d): redone %d retried %d, operator call %.0f us (host-inclusive)"
                  % (T, N, "cat-mod" if mods else "plain  ", nlong, frac, int(Ls.max()), g, r, dt * 1e6), flush=True)
