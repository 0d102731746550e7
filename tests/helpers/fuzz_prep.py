#!/usr/bin/env python
"""Random-parameter sweep of the data-preparation kernels against their oracles (GPU box; test
infrastructure): chunk batches (chunk length, batch size, filters, direction, raw / standardised
signal, cat-mod labels, device-drawn candidates) and remapping (blocks, bases, localpen, nbase).

    python -m tests.helpers.fuzz_prep [--cases 20] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import chunks as oc, remap as orm  # noqa: E402
from taiyaki_amd import flipflop_remap as fr, mapped_signal as ms, synth  # noqa: E402


def chunk_case(k, rng, dev):
    nreads = int(rng.randint(3, 60))
    mod = bool(rng.randint(2))
    reads = synth.mapped_reads(nreads, 9000 + k, nlabel=6 if mod else 4, mean_reflen=int(rng.randint(40, 900)),
                               long_dwell_prob=float(rng.choice([0.0, 0.001, 0.01])))
    store = ms.MappedSignalStore(reads, dev)
    T = int(rng.randint(20, 3000))
    nwant = int(rng.randint(1, 200))
    stride = int(rng.choice([1, 2, 5, 8]))
    minpass = float(rng.choice([0.1, 0.25, 0.5, 1.0]))
    torch.manual_seed(k)
    fp = store.sample_filter_parameters(int(rng.randint(5, 300)), T, float(rng.uniform(0.5, 4.0)),
                                        float(rng.uniform(2.0, 12.0)), minpass, stride,
                                        float(rng.choice([0.5, 1.0, 1.1, 1.5])))
    if not np.isfinite(fp.median_meandwell):                    # nothing long enough: no filtering
        fp = ms.FILTER_PARAMETERS(fp.filter_mean_dwell, fp.filter_max_dwell, minpass, None, None, None, None)
    rev, std = bool(rng.randint(2)), bool(rng.randint(2))
    kw = dict(can_labels=[0, 1, 2, 3, 0, 1], mod_labels=[0, 0, 0, 0, 1, 1]) if mod else {}
    b = store.sample_chunks(nwant, T, fp, standardize=std, reverse=rev, max_bases_per_chunk=T + 8, **kw)
    cr, dacstart = b.cand_read.cpu().numpy(), b.dacstart.cpu().numpy()
    cands = []
    for i, rn in enumerate(cr):
        spare = int(store.mapped[rn, 1]) - int(store.mapped[rn, 0]) - T
        cands.append((int(rn), int(dacstart[i] - store.mapped[rn, 0]) if spare > 0 else 0))
    want, counts, attempts = oc.sample_chunks(reads, nwant, T, dict(fp._asdict()), cands, standardize=std)
    ok = b.rejections() == {q: v for q, v in counts.items() if v} and b.attempts == attempts and \
        b.naccepted == len(want)
    if ok and want:
        indata, seqs, seqlens, mods = oc.assemble_batch(want, 4, reverse=rev, can_labels=kw.get("can_labels"),
                                                        mod_labels=kw.get("mod_labels"))
        got = b.trimmed()
        ok = (np.array_equal(got[0].cpu().numpy().view(np.uint32), indata.view(np.uint32)) and
              np.array_equal(got[1].cpu().numpy(), seqs) and np.array_equal(got[2].cpu().numpy(), seqlens) and
              (mods is None or np.array_equal(got[3].cpu().numpy(), mods)))
    return ok, "chunks reads=%d T=%d want=%d accepted=%d stride=%d rev=%d std=%d mod=%d" % (
        nreads, T, nwant, b.naccepted, stride, rev, std, mod)


def remap_case(k, rng):
    nb = int(rng.choice([2, 4]))
    T, M = int(rng.randint(1, 3000)), int(rng.randint(1, 2600))
    pen = float(rng.choice([1e30, 0.3, 2.0, 6.0, -0.5]))
    sc = synth.scores(T, 1, 2 * nb * (nb + 1), 9500 + k)[:, 0, :]
    if rng.randint(2):
        sc = np.round(sc).astype(np.float32)                # ties
    bases = rng.randint(0, nb, size=M)
    ws, wp = orm.flipflop_remap(sc, bases, nb, localpen=pen)
    s, p = fr.flipflop_remap(sc, bases, alphabet="ACGT"[:nb], localpen=pen)
    return (s == ws and np.array_equal(p, wp)), "remap T=%d M=%d nb=%d pen=%g" % (T, M, nb, pen)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    dev = torch.device("cuda:0")
    bad = 0
    for k in range(args.cases):
        for ok, what in (chunk_case(k, rng, dev), remap_case(k, rng)):
            bad += not ok
            print("%s %s" % ("ok  " if ok else "FAIL", what), flush=True)
    print("fuzz: %d cases, %d failures" % (2 * args.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
