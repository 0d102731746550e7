#!/usr/bin/env python
"""Generate tests/golden/remap_small.npz with the GENUINE reference (this container only):
taiyaki.flipflop_remap.flipflop_remap on scores from taiyaki_amd.synth, global (default localpen)
and glocal.  The float32 scores are handed to the reference as float64 (same values): under the
numpy 2 in this image a float32 scalar plus a Python float stays float32, which would turn the
reference's start / end state scores into float32 accumulators -- the numpy 1.x it was released
against promotes them to float64, and so does a float64 input on any numpy.  For the same reason
the module's `np.unpackbits` is widened to int64 (see NumpyOneInts).  Only seeds and OUTPUTS
(score, path) are stored.

    python tests/golden/make_golden_remap.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden  # noqa: E402
from tests.golden.cases import REMAP_SMALL, remap_inputs  # noqa: E402


def main():
    make_golden.build_reference()
    from taiyaki import flipflop_remap

    class NumpyOneInts:
        """numpy seen by the reference module, with unpackbits widened to int64: numpy 2 refuses
        `m -= np.uint8(move)` for a Python int m > 255 (flipflop_remap.py:84-85), numpy 1.x
        promoted it.  No algorithmic change."""

        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def unpackbits(a):
            return np.unpackbits(a).astype(np.int64)
    flipflop_remap.np = NumpyOneInts()
    out = {}
    for name, spec in REMAP_SMALL.items():
        scores, bases = remap_inputs(spec)
        alphabet = "ACGTZYXW"[:spec["nbase"]]
        seq = "".join(alphabet[b] for b in bases)
        kw = {} if spec["localpen"] is None else dict(localpen=spec["localpen"])
        score, path = flipflop_remap.flipflop_remap(scores.astype(np.float64), seq, alphabet=alphabet, **kw)
        out[name + "/score"] = np.float64(score)
        out[name + "/path"] = np.asarray(path, dtype=np.int64)
        print(name, "score", score, "clipped", int((path < 0).sum()), "of", len(path))
    path = os.path.join(HERE, "remap_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
