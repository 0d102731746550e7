#!/usr/bin/env python
"""Generate tests/golden/chunks_small.npz with the GENUINE reference (this container only):
signal_mapping.SignalMapping.get_chunk_with_sample_length, Chunk.apply_filters,
chunk_selection.sample_filter_parameters / sample_chunks and the batch stacking of
bin/train_flipflop.py:prepare_random_batches, on reads from taiyaki_amd.synth.mapped_reads.
The reference draws (read number, start sample) from numpy's global generator; the draws are
recorded by wrapping np.random.randint so that the fixture holds the candidate list next to
the expected outputs.  Only seeds and OUTPUTS are stored.

    python tests/golden/make_golden_chunks.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden  # noqa: E402
from tests.golden.cases import CHUNKS_SMALL, chunk_reads  # noqa: E402


def main():
    make_golden.build_reference()
    from taiyaki import chunk_selection, flipflopfings, signal_mapping
    out = {}
    for name, spec in CHUNKS_SMALL.items():
        reads = chunk_reads(spec)
        sms = [signal_mapping.SignalMapping(
            r["Ref_to_signal"], r["Reference"], signalstart=0,
            **{k: r[k] for k in ("shift_frompA", "scale_frompA", "range", "offset", "digitisation",
                                 "read_id", "Dacs")}) for r in reads]
        assert all(sm.check() == sm.pass_str for sm in sms)
        draws = []
        real_randint = np.random.randint

        def recording_randint(high):
            v = real_randint(high)
            draws.append((int(high), int(v)))
            return v
        np.random.randint = recording_randint
        try:
            np.random.seed(spec["seed"])
            fp = chunk_selection.sample_filter_parameters(
                sms, spec["nsample"], spec["chunk_len"], spec["filter_mean_dwell"],
                spec["filter_max_dwell"], spec["min_pass"], spec["stride"], spec["path_buffer"])
            ndraw_fp = len(draws)
            chunks, rej = chunk_selection.sample_chunks(
                sms, spec["nwant"], spec["chunk_len"], fp, standardize=spec["standardize"])
        finally:
            np.random.randint = real_randint
        out[name + "/median_mad"] = np.array([fp.median_meandwell, fp.mad_meandwell])
        out[name + "/draws_filter"] = np.array(draws[:ndraw_fp], dtype=np.int64).reshape(-1, 2)
        out[name + "/draws_batch"] = np.array(draws[ndraw_fp:], dtype=np.int64).reshape(-1, 2)
        out[name + "/rejections"] = np.array(
            [rej.get(k, 0) for k in ("pass", "emptysequence", "emptysignal", "tooshort", "nullmapping",
                                     "pathbuffer", "meandwell", "maxdwell")], dtype=np.int64)
        print(name, "accepted", len(chunks), "of", spec["nwant"], dict(rej), "median/mad", fp.median_meandwell,
              fp.mad_meandwell)
        assert len(chunks) > 0
        ids = [r["read_id"] for r in reads]
        out[name + "/chunk_read"] = np.array([ids.index(c.read_id) for c in chunks], dtype=np.int64)
        out[name + "/chunk_start"] = np.array([c.start_sample for c in chunks], dtype=np.int64)
        out[name + "/chunk_maxdwell"] = np.array([c.max_dwell for c in chunks], dtype=np.int64)
        # the stacking of prepare_random_batches (bin/train_flipflop.py:103-140), both directions
        for tag, revop in (("fwd", np.array), ("rev", np.flip)):
            cur = np.vstack([revop(c.current) for c in chunks]).T
            out[name + "/indata_" + tag] = cur.astype(np.float32)[:, :, None]
            seqs, seqlens, mods = [], [], []
            for c in chunks:
                lab = revop(c.sequence)
                seqlens.append(len(lab))
                if spec["mod"]:
                    mods.append(np.asarray(spec["mod_labels"])[lab])
                    lab = np.asarray(spec["can_labels"])[lab]
                seqs.append(flipflopfings.flipflop_code(np.ascontiguousarray(lab), spec["ncan"]))
            out[name + "/seqs_" + tag] = np.concatenate(seqs).astype(np.int64)
            out[name + "/seqlens_" + tag] = np.array(seqlens, dtype=np.int64)
            if spec["mod"]:
                out[name + "/modcats_" + tag] = np.concatenate(mods).astype(np.int64)
    path = os.path.join(HERE, "chunks_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
