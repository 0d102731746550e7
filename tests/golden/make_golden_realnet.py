#!/usr/bin/env python
"""Generate tests/golden/realnet.npz: a TRAINED network's scores on REAL reads, with the genuine
reference's answers on them (this container only).

    python tests/golden/make_golden_realnet.py

Every other fixture of this directory holds iid or synthetic "confident" scores.  The reference ships
one trained flip-flop network and seven real mapped reads:
    /root/reference/models/mGru_flipflop_remapping_model_r9_DNA.checkpoint      (configs[0]'s family)
    /root/reference/test/data/mapped_signal_file/mapped_reads_{0,1}.hdf5
This script (nothing of the reference is copied: only NUMBERS travel)
 1. builds the scratch copy of the reference (make_golden.build_reference) and unpickles the network
    with one load-only adaptation: the checkpoint was written by a torch whose GRU modules carry no
    `_flat_weights` list, which torch 2.10's `RNNBase.__setstate__` indexes -- the list is rebuilt
    from the module's own `_all_weights` names (no weight is touched);
 2. reads the seven reads with this repository's HDF5 reader (pinned against the same files by
    tests/test_hdf5_reader.py), wraps them in the reference's `SignalMapping` and cuts chunks with
    the reference's own `sample_filter_parameters` / `sample_chunks` (chunk_selection.py:29-131) and
    the stacking of `prepare_random_batches` (bin/train_flipflop.py:78-142), chunk_len 2000
    (configs[0]'s shape; this network strides by 4: T = 500);
 3. runs the network (CPU, fp32) and stores its (T, N, 40) output;
 4. evaluates on it, with the GENUINE reference: `crf_flipflop_loss` at sharpen 1.0 and 2.0 (loss,
    gradient checksums), `flipflop_logpartition` (value, gradient checksums), `_flipflop_viterbi`
    (fwd's last column, traceback sums, paths), `flipflop_make_trans` row sums,
    `flipflop_remap` (path + score) and `decodeutil.beamsearch` (reference C) of two reads;
 5. the cat-mod loss on the same canonical scores with synthetic modification columns (step 6 below);
 5b. a second case, "fast": chunks of 3400 samples resampled to 2000 (linear interpolation), the
    labels of the whole 3400-sample stretch kept: reads of L ~ 0.75 T, where the band of the
    sequence lattice is narrow, on scores a network produced.
"""
import os
import sys
import warnings
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden  # noqa: E402
from tests.golden.cases import grad_checksums  # noqa: E402

CHECKPOINT = "/root/reference/models/mGru_flipflop_remapping_model_r9_DNA.checkpoint"
READ_FILES = ["/root/reference/test/data/mapped_signal_file/mapped_reads_%d.hdf5" % k for k in (0, 1)]
CHUNK_LEN, FAST_LEN = 2000, 3400
NCHUNK = dict(real=32, fast=16)       # fp32 scores do not compress: 2.6 + 1.3 MB


def load_network():
    import torch
    orig = torch.nn.RNNBase.__setstate__

    def setstate(self, d):
        if "_flat_weights" in d:
            return orig(self, d)
        torch.nn.Module.__setstate__(self, d)
        self.proj_size = 0
        names = [n for ws in self._all_weights for n in ws]
        self._flat_weights_names = names
        self._flat_weights = [getattr(self, n) for n in names]
        self._flat_weight_refs = [weakref.ref(w) for w in self._flat_weights]
    torch.nn.RNNBase.__setstate__ = setstate
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = torch.load(CHECKPOINT, map_location="cpu", weights_only=False)
    finally:
        torch.nn.RNNBase.__setstate__ = orig
    assert net.metadata["version"] == 3 and not net.metadata["reverse"] and net.metadata["standardize"]
    return net.eval()


def cut_chunks(sms, chunk_len, nwant, seed):
    """The reference's producer: filter parameters from 200 sampled chunks, then a batch."""
    from taiyaki import chunk_selection
    np.random.seed(seed)
    fp = chunk_selection.sample_filter_parameters(sms, 200, chunk_len, 10.0, 10.0, 0.5, 4, 1.1)
    chunks, rej = chunk_selection.sample_chunks(sms, nwant, chunk_len, fp, standardize=True)
    print("chunk_len", chunk_len, "accepted", len(chunks), dict(rej))
    return chunks


def main():
    make_golden.build_reference()
    import torch
    from taiyaki import ctc, decode, flipflop_remap, flipflopfings, layers, signal_mapping
    from taiyaki_amd import hdf5_lite
    from oracle import beam
    torch.set_num_threads(8)
    net = load_network()

    reads = []
    for path in READ_FILES:
        info, rs = hdf5_lite.read_mapped_signal_file(path)
        assert info["alphabet"] == "ACGT", info
        reads += rs
    sms = [signal_mapping.SignalMapping(
        r["Ref_to_signal"], r["Reference"], signalstart=0,
        **{k: r[k] for k in ("shift_frompA", "scale_frompA", "range", "offset", "digitisation",
                             "read_id", "Dacs")}) for r in reads]
    assert all(sm.check() == sm.pass_str for sm in sms)
    print(len(sms), "reads,", sum(len(r["Dacs"]) for r in reads), "samples")

    def t(x, dtype=None):
        return torch.tensor(np.asarray(x), dtype=dtype)

    class NumpyOneInts:                   # see make_golden_remap.py (numpy 2 vs the uint8 arithmetic)
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def unpackbits(a):
            return np.unpackbits(a).astype(np.int64)
    flipflop_remap.np = NumpyOneInts()

    out = {}
    for tag, chunk_len, seed in (("real", CHUNK_LEN, 1), ("fast", FAST_LEN, 2)):
        chunks = cut_chunks(sms, chunk_len, NCHUNK[tag], seed)
        assert len(chunks) == NCHUNK[tag]
        cur = np.vstack([np.array(c.current) for c in chunks]).T.astype(np.float32)     # (chunk_len, N)
        if chunk_len != CHUNK_LEN:        # a read that runs 1.7x faster through the pore
            x_new = np.linspace(0.0, chunk_len - 1.0, CHUNK_LEN)
            cur = np.stack([np.interp(x_new, np.arange(chunk_len), cur[:, n]) for n in range(cur.shape[1])],
                           axis=1).astype(np.float32)
        seqs = [flipflopfings.flipflop_code(np.ascontiguousarray(np.array(c.sequence)), 4) for c in chunks]
        bases = [np.array(c.sequence, dtype=np.int8) for c in chunks]
        seqlens = np.array([len(s) for s in seqs], dtype=np.int64)
        seqs = np.concatenate(seqs).astype(np.int64)
        with torch.no_grad():
            scores = net(t(cur).unsqueeze(2)).numpy().astype(np.float32)
        T, N, S = scores.shape
        assert (T, N, S) == (CHUNK_LEN // 4, NCHUNK[tag], 40)
        print(tag, "scores", scores.shape, "range", scores.min(), scores.max(), "L/T", (seqlens / T).round(2))
        out[tag + "/scores"] = scores
        out[tag + "/seqs"] = seqs.astype(np.int8)
        out[tag + "/bases"] = np.concatenate(bases)
        out[tag + "/seqlens"] = seqlens
        for sharp in (1.0, 2.0):
            x = t(scores).requires_grad_()
            loss = ctc.crf_flipflop_loss(x, t(seqs), t(seqlens), sharp)
            loss.sum().backward()
            k = "%s/crf_s%d" % (tag, int(sharp))
            out[k + "_loss"] = loss.detach().numpy()
            for name, v in grad_checksums(x.grad.numpy()).items():
                out[k + "_grad_" + name] = v
            print(tag, "sharpen", sharp, "loss", loss.detach().numpy()[:4])
        x = t(scores).requires_grad_()
        lz = layers.flipflop_logpartition(x)
        lz.sum().backward()
        out[tag + "/logz"] = lz.detach().numpy()
        for name, v in grad_checksums(x.grad.numpy()).items():
            out[tag + "/logz_grad_" + name] = v
        # calculate_loss's assembly (bin/train_flipflop.py:172-182)
        out[tag + "/lossvector"] = out[tag + "/crf_s1_loss"] + out[tag + "/logz"] / T
        print(tag, "logz", out[tag + "/logz"][:4], "(this model version emits 5 tanh scores; calculate_loss adds logZ / T)")
        fwd, tb, path = decode.flipflop_viterbi(t(scores), _never_use_cupy=True)
        out[tag + "/vit_path"] = path.numpy().astype(np.int8)
        out[tag + "/vit_fwd_last"] = fwd[-1].numpy()
        out[tag + "/vit_tb_sum"] = tb.numpy().sum(axis=(0, 2))
        trans = decode.flipflop_make_trans(t(scores), _never_use_cupy=True)
        out[tag + "/trans_rowsum"] = trans.sum(dim=2).numpy()
        for n in (0, 1):
            lo = int(seqlens[:n].sum())
            seq = "".join("ACGT"[b] for b in np.concatenate(bases)[lo:lo + seqlens[n]])
            score, rpath = flipflop_remap.flipflop_remap(scores[:, n, :].astype(np.float64), seq, alphabet="ACGT")
            out["%s/remap%d_score" % (tag, n)] = np.float64(score)
            out["%s/remap%d_path" % (tag, n)] = np.asarray(rpath, dtype=np.int64)
            bseq, bscore, _ = beam.ref_beamsearch(scores[:, n, :], 0.0, 5, True)
            out["%s/beam%d_seq" % (tag, n)] = np.asarray(bseq, dtype=np.int8)
            out["%s/beam%d_score" % (tag, n)] = np.float32(bscore)
            print(tag, "read", n, "remap score", score, "beam", len(bseq), "bases, score", bscore,
                  "true length", seqlens[n])
    # 6. the cat-mod loss on the same canonical scores (cases.realnet_catmod_inputs: synthetic modification
    #    columns as the cat-mod layer emits them), sharpened as bin/train_flipflop.py:161-170 does
    from tests.golden.cases import realnet_catmod_inputs
    for tag in ("real", "fast"):
        inp = realnet_catmod_inputs(out, tag)
        for sharp in (1.0, 2.0):
            x = t(inp["scores"]).requires_grad_()
            loss = ctc.cat_mod_flipflop_loss(x, t(inp["seqs"]), t(inp["seqlens"], torch.int64), t(inp["mod_cats"]),
                                             inp["can_mods_offsets"], inp["mod_cat_weights"], sharp)
            loss.sum().backward()
            k = "%s/catmod_s%d" % (tag, int(sharp))
            out[k + "_loss"] = loss.detach().numpy()
            for name, v in grad_checksums(x.grad.numpy()).items():
                out[k + "_grad_" + name] = v
            print(tag, "cat-mod sharpen", sharp, "loss", loss.detach().numpy()[:4])
    path = os.path.join(HERE, "realnet.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
