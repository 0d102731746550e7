#!/usr/bin/env python
"""Generate tests/golden/basecall_small.npz with the GENUINE reference (this container only):
decode.flipflop_make_trans, decode.flipflop_viterbi, qscores.errprobs_from_trans,
basecall_helpers.chunk_read / stitch_chunks, qscores.path_errprobs_to_qstring evaluated on
inputs from taiyaki_amd.synth.  Run tests/golden/make_golden.py first (it builds the scratch
copy of the reference under /tmp); only inputs' seeds and expected OUTPUTS are stored.

    python tests/golden/make_golden_basecall.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden  # noqa: E402
from tests.golden.cases import BASECALL_SMALL, basecall_scores  # noqa: E402


def main():
    make_golden.build_reference()
    import torch
    from taiyaki import basecall_helpers, decode, qscores
    if not hasattr(np.ndarray, "tostring"):         # numpy 2 dropped the alias the reference uses
        def qchar_from_qscore(score, zerochar=33):
            return (np.array(score) + zerochar + 0.5).astype(np.int8).tobytes().decode("ascii")
        qscores.qchar_from_qscore = qchar_from_qscore
    out = {}
    for name, spec in BASECALL_SMALL.items():
        scores = torch.tensor(basecall_scores(spec))
        T, N, _ = scores.shape
        trans = decode.flipflop_make_trans(scores, _never_use_cupy=True).detach()
        _, _, path = decode.flipflop_viterbi(scores, _never_use_cupy=True)
        err = qscores.errprobs_from_trans(trans, path)
        out[name + "/trans_sum"] = trans.sum(dim=2).numpy()
        out[name + "/path"] = path.numpy()
        out[name + "/errprobs"] = err.numpy()
        # chunk geometry from the reference's own chunker: N overlapping chunks of T*stride samples
        stride, chunk = spec["stride"], T * spec["stride"]
        siglen = chunk + (N - 1) * (chunk - spec["overlap"]) - spec["ragged"]
        _, starts, ends = basecall_helpers.chunk_read(np.zeros(siglen, dtype="f4"), chunk, spec["overlap"])
        assert len(starts) == N, (len(starts), N)
        out[name + "/chunk_starts"], out[name + "/chunk_ends"] = starts, ends
        spath = basecall_helpers.stitch_chunks(path, starts, ends, stride)
        serr = basecall_helpers.stitch_chunks(err, starts, ends, stride)
        out[name + "/stitched_path"] = spath.numpy()
        out[name + "/stitched_errprobs"] = serr.numpy()
        out[name + "/stitched_path_ps"] = basecall_helpers.stitch_chunks(
            path, starts, ends, stride, path_stitching=True).numpy()
        q = qscores.path_errprobs_to_qstring(serr, spath.numpy(), 0.9, 0.3)
        out[name + "/qstring"] = np.frombuffer(q.encode("ascii"), dtype=np.uint8)
    # rolling median + MAD thresholds of the gradient clipper (maths.py:138-195)
    from taiyaki import maths
    rng = np.random.RandomState(7)
    vals = np.abs(rng.standard_normal((14, 5))).astype(np.float32) * np.float32(3.0)
    out["rollingmad/vals"] = vals
    for tag, n_mads in (("m0", 0), ("m15", 1.5)):
        rm = maths.RollingMAD(5, n_mads=n_mads, window=6)
        ths = []
        for v in vals:
            th = rm.update(list(v))
            ths.append(np.full(5, np.nan, dtype=np.float64) if th is None else np.asarray(th, dtype=np.float64))
        out["rollingmad/thresh_" + tag] = np.stack(ths)
    np.savez_compressed(os.path.join(HERE, "basecall_small.npz"), **out)
    print("wrote basecall_small.npz:", sorted(out)[:6], "...")


if __name__ == "__main__":
    main()
