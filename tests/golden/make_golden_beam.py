#!/usr/bin/env python
"""Golden vectors for the hash beam search from the GENUINE reference C (this container only):
oracle/Makefile compiles taiyaki/decodeutil/{c_hashdecode,c_flipflopfwdbwd,fasthash,yastring}.c
where they lie under /root/reference into oracle/_ref/libref_decodeutil.so; this script runs
`flipflop_backward` + `flipflop_beamsearch` through it with the wrapper logic of
decodeutil.pyx:36-51 on seeded synthetic scores and stores the decoded sequences and scores.

    python tests/golden/make_golden_beam.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import beam  # noqa: E402
from taiyaki_amd import synth  # noqa: E402

# name: (T, seed, scale, beam_width, beam_cut, guided)
CASES = dict(t40w5=(40, 11, 0.7, 5, 0.0, True), t300w5=(300, 12, 0.7, 5, 0.0, True),
             t300w5_sharp=(300, 13, 1.5, 5, 0.0, True), t200w12=(200, 14, 0.7, 12, 0.0, True),
             t250w3_unguided=(250, 15, 0.7, 3, 0.0, False), t250w5_cut=(250, 16, 0.7, 5, 0.02, True),
             t120w1=(120, 17, 1.0, 1, 0.0, True), t800w5=(800, 18, 0.9, 5, 0.0, True))
# tie-heavy cases (scores on a grid of 0.5, tests/test_beamsearch.py:quantised_scores): (T, seed, width, cut, guided)
QUANTISED = dict(q150w5=(150, 21, 5, 0.0, True), q120w12=(120, 22, 12, 0.0, True), q90w7_cut=(90, 23, 7, 0.05, True))


def case_scores(spec):
    T, seed, scale = spec[:3]
    return (synth.scores(T, 1, 40, seed)[:, 0, :] * np.float32(scale)).astype(np.float32)


def main():
    assert beam.ref_available(), "build oracle/_ref first (make -C oracle)"
    out = {}
    for name, spec in CASES.items():
        seq, score, _ = beam.ref_beamsearch(case_scores(spec), spec[4], spec[3], spec[5])
        out[name + "/seq"] = seq
        out[name + "/score"] = np.float32(score)
        print(name, len(seq), score)
    from tests.test_beamsearch import quantised_scores
    for name, (T, seed, w, cut, g) in QUANTISED.items():
        seq, score, _ = beam.ref_beamsearch(quantised_scores(T, seed), cut, w, g)
        out[name + "/seq"] = seq
        out[name + "/score"] = np.float32(score)
        print(name, len(seq), score)
    np.savez_compressed(os.path.join(HERE, "beam_small.npz"), **out)


if __name__ == "__main__":
    main()
