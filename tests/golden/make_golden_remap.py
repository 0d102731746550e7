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
    from taiyaki import flipflop_remap, signal_mapping

    class SigStub:
        def __init__(self, dacs, signalstart):
            self.untrimmed_dacs, self.signalstart = dacs, signalstart
            self.shift_from_pA, self.scale_from_pA, self.range, self.offset = 0.0, 1.0, 1.0, 0.0
            self.digitisation, self.read_id = 1.0, "stub"

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
        # the mapping the reference derives from that path (prepare_mapping_funcs.py:95-97):
        # SignalMapping.from_remapping_path with a stand-in for the Signal object (taiyaki.signal
        # needs ont_fast5_api; only these attributes are read)
        for stride, start, extra in ((1, 0, 0), (5, 3, 7), (2, 40, 0)):
            sig = SigStub(np.zeros(len(path) * stride + start + extra, dtype=np.int16), start)
            sm = signal_mapping.SignalMapping.from_remapping_path(
                np.asarray(path), np.asarray(bases, dtype=np.int16), stride, sig)
            out["%s/rts_s%d_o%d_e%d" % (name, stride, start, extra)] = sm.Ref_to_signal.astype(np.int32)
        print(name, "score", score, "clipped", int((path < 0).sum()), "of", len(path))
    path = os.path.join(HERE, "remap_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
