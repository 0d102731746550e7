#!/usr/bin/env python
"""Run every parity comparison on the GPU and PRINT the error figures without
asserting -- one round trip to the GPU box gives a complete picture.
    python tests/gpu_diag.py [--full]
"""
import os
import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from tests import parity  # noqa: E402
from tests.conftest import load_golden  # noqa: E402
from tests.golden import cases  # noqa: E402


def show(tag, r):
    keys = [k for k in r if not isinstance(r[k], np.ndarray)]
    print("%-28s " % tag + "  ".join("%s=%s" % (k, ("%.3g" % r[k]) if isinstance(r[k], float) else r[k])
                                      for k in keys), flush=True)


def guarded(tag, fn):
    try:
        t0 = time.time()
        r = fn()
        show(tag, r)
        print("    (%.2fs)" % (time.time() - t0), flush=True)
        return r
    except Exception:
        print("%-28s EXCEPTION" % tag)
        traceback.print_exc()
        return None


def main():
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0), flush=True)
    oracle.build()
    for name, spec in cases.LOGZ_SMALL.items():
        sc = cases.logz_inputs(spec)
        guarded("logz/" + name, lambda: parity.compare_logz(oracle, sc, dev))
        guarded("viterbi/" + name, lambda: parity.compare_viterbi(oracle, sc, dev))
    for name, spec in cases.CRF_SMALL.items():
        inp = cases.crf_inputs(spec)
        guarded("crf/" + name, lambda: parity.compare_crf(oracle, inp, spec["sharp"], dev))
    for name, spec in cases.CATMOD_SMALL.items():
        inp = cases.crf_inputs(spec, cases.NMODS)
        guarded("catmod/" + name, lambda: parity.compare_crf(oracle, inp, spec["sharp"], dev))
    if "--full" in sys.argv:
        gold = load_golden("fullsize.npz")
        for name in cases.FULLSIZE:
            inp = parity.fullsize_inputs(name)

            def crf_full():
                loss, grad = parity.run_crf(inp, 1.0, dev)
                cs = cases.grad_checksums(grad)
                return dict(loss_rel=parity.rel_err(loss, gold[name + "/loss"]),
                            sample_abs=parity.abs_err(cs["sample"], gold[name + "/grad_sample"]),
                            sum_rel=parity.rel_err(cs["sum"], gold[name + "/grad_sum"]))

            def logz_full():
                sc40 = np.ascontiguousarray(inp["scores"][:, :, :40])
                lz, g = parity.run_logz(sc40, dev)
                cs = cases.grad_checksums(g)
                _, _, path = parity.run_viterbi(sc40, dev)
                return dict(logz_rel=parity.rel_err(lz, gold[name + "/logz"]),
                            sample_abs=parity.abs_err(cs["sample"], gold[name + "/lgrad_sample"]),
                            rowsum=float(np.abs(g.sum(axis=2) - 1).max()),
                            path_hash_mismatch=int((parity.path_hash(path) !=
                                                    gold[name + "/path_hash"]).sum()))
            guarded("full-crf/" + name, crf_full)
            guarded("full-logz/" + name, logz_full)


if __name__ == "__main__":
    main()
