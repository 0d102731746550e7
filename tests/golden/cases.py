"""Case list shared by make_golden.py (which runs the genuine reference) and the
parity tests.  Inputs are regenerated from taiyaki_amd.synth; the fixtures hold
only the reference's OUTPUTS (plus a few small stored inputs)."""
import numpy as np

from taiyaki_amd import synth

# name -> dict(T, N, seed, sharp, seqlens=None|list, nbase)
CRF_SMALL = {
    "t4n1": dict(T=4, N=1, seed=11, sharp=1.0, seqlens=[3]),
    "t7n2": dict(T=7, N=2, seed=12, sharp=1.0, seqlens=[6, 2]),
    "t7n2_len1": dict(T=7, N=2, seed=13, sharp=1.0, seqlens=[1, 5]),
    "t50n4": dict(T=50, N=4, seed=14, sharp=1.0, seqlens=[20, 45, 1, 33]),
    "t50n4_sharp": dict(T=50, N=4, seed=15, sharp=2.5, seqlens=[25, 40, 7, 12]),
    "t50n3_zero_last": dict(T=50, N=3, seed=16, sharp=1.0, seqlens=[30, 17, 0]),
    "t64n8": dict(T=64, N=8, seed=17, sharp=1.0, seqlens=None),
    "t200n8": dict(T=200, N=8, seed=18, sharp=1.0, seqlens=None),
    "t130n5_long": dict(T=130, N=5, seed=19, sharp=1.0,
                        seqlens=[118, 65, 64, 63, 129]),
    "t300n3_wide": dict(T=300, N=3, seed=20, sharp=1.0, seqlens=[270, 257, 129]),
}

CATMOD_SMALL = {
    "t7n2": dict(T=7, N=2, seed=31, sharp=1.0, seqlens=[6, 3]),
    "t50n4": dict(T=50, N=4, seed=32, sharp=1.0, seqlens=[20, 45, 2, 33]),
    "t50n4_sharp": dict(T=50, N=4, seed=33, sharp=2.5, seqlens=[25, 40, 7, 12]),
    "t200n8": dict(T=200, N=8, seed=34, sharp=1.0, seqlens=None),
}

# (T, N, nbase, seed)
LOGZ_SMALL = {
    "t1n1": dict(T=1, N=1, nbase=4, seed=41),
    "t7n3_nb2": dict(T=7, N=3, nbase=2, seed=42),
    "t50n5": dict(T=50, N=5, nbase=4, seed=43),
    "t200n8": dict(T=200, N=8, nbase=4, seed=44),
    "t333n70": dict(T=333, N=70, nbase=4, seed=45),
    "t100n4_nb3": dict(T=100, N=4, nbase=3, seed=46),
}

# basecall-side consumers: N chunks of T blocks that overlap by `overlap` samples;
# `ragged` shortens the signal so that the last chunk overlaps its neighbour by more
BASECALL_SMALL = {
    "t60n5": dict(T=60, N=5, nbase=4, seed=61, stride=5, overlap=100, ragged=0),
    "t200n70": dict(T=200, N=70, nbase=4, seed=62, stride=5, overlap=100, ragged=130),
    "t40n1": dict(T=40, N=1, nbase=4, seed=63, stride=2, overlap=20, ragged=0),
    "t90n3_nb2": dict(T=90, N=3, nbase=2, seed=64, stride=3, overlap=60, ragged=31),
}


def basecall_scores(spec):
    nb = spec["nbase"]
    return synth.scores(spec["T"], spec["N"], 2 * nb * (nb + 1), spec["seed"])


# BASELINE.json configs at full size: only per-read scalars + checksums are kept
FULLSIZE = {
    "cfg1": dict(T=1000, N=64, seed=4, mods=None),      # train_abinitio: chunk 2000, stride 2, batch 64
    "cfg2": dict(T=800, N=128, seed=1, mods=None),
    "cfg4": dict(T=800, N=128, seed=2, mods=(1, 1, 0, 0)),
    # cfg 4 on PRODUCER-LIKE inputs (round 5): the modification columns are what GlobalNormFlipFlopCatMod.forward
    # emits (layers.py:1616-1640: per-base log-softmax; synth.normalise_mod_columns) instead of cfg4's raw
    # U(-5, 5) logits x 8, most of whose reads the linear path disowns -- here the LINEAR cat-mod kernel at
    # T 800 / N 128 is what gets compared with the genuine reference (gate count asserted 0)
    "cfg4_lsm": dict(T=800, N=128, seed=2, mods=(1, 1, 0, 0), lsm=0.2),
    "cfg5": dict(T=1600, N=64, seed=3, mods=None),
    "rowK": dict(T=4000, N=256, seed=1, mods=None),
}

NMODS = (1, 1, 0, 0)        # ACGTZY: 6mA on A, 5mC on C


def crf_inputs(spec, mods=None):
    seqlens = spec.get("seqlens")
    inp = synth.crf_case(spec["T"], spec["N"], spec["seed"],
                         nbase=spec.get("nbase", 4), nmods_per_base=mods,
                         seqlens=seqlens)
    if mods is not None and spec.get("lsm"):
        synth.normalise_mod_columns(inp, logit_scale=spec["lsm"])
    return inp


def logz_inputs(spec):
    nb = spec["nbase"]
    return synth.scores(spec["T"], spec["N"], 2 * nb * (nb + 1), spec["seed"])


def grad_checksums(grad):
    """Size-independent digest of a (T,N,S) gradient: per-read sum, sum of squares
    (float64) and a strided sample of raw elements."""
    g = np.asarray(grad, dtype=np.float64)
    flat = np.asarray(grad).reshape(-1)
    idx = np.arange(0, flat.size, max(1, flat.size // 4096))[:4096]
    return dict(sum=g.sum(axis=(0, 2)), sumsq=(g * g).sum(axis=(0, 2)),
                sample_idx=idx, sample=flat[idx].copy())


# training-chunk extraction: reads from synth.mapped_reads, the reference's
# sample_filter_parameters + sample_chunks on them
_CH = dict(filter_mean_dwell=3.0, filter_max_dwell=10.0, min_pass=0.5, stride=5, path_buffer=1.1,
           standardize=True, mod=False, ncan=4, nlabel=4, can_labels=None, mod_labels=None)
CHUNKS_SMALL = {
    "r40c600": dict(_CH, nreads=40, seed=71, chunk_len=600, nsample=60, nwant=32),
    "r25c2000": dict(_CH, nreads=25, seed=72, chunk_len=2000, nsample=40, nwant=16),
    "r30c300_tight": dict(_CH, nreads=30, seed=73, chunk_len=300, nsample=50, nwant=48, stride=8,
                          filter_mean_dwell=1.0, filter_max_dwell=4.0, min_pass=0.2),
    "r20c500_raw": dict(_CH, nreads=20, seed=74, chunk_len=500, nsample=30, nwant=20,
                        standardize=False),
    # ACGTZY: labels 4 (6mA) and 5 (5mC) are modified A and C
    "r30c800_mod": dict(_CH, nreads=30, seed=75, chunk_len=800, nsample=40, nwant=24, mod=True,
                        nlabel=6, can_labels=[0, 1, 2, 3, 0, 1], mod_labels=[0, 0, 0, 0, 1, 1]),
    "r6c700_starved": dict(_CH, nreads=6, seed=76, chunk_len=700, nsample=12, nwant=64,
                            filter_max_dwell=6.0, min_pass=0.25),
}


def chunk_reads(spec):
    return synth.mapped_reads(spec["nreads"], spec["seed"], nlabel=spec["nlabel"])


# flipflop_remap: (T, K) scores, a base sequence of length M, localpen (None = the
# reference's default LARGE_VAL = global mapping)
REMAP_SMALL = {
    "t30m8": dict(T=30, M=8, nbase=4, seed=81, localpen=None, scale=1.0),
    "t200m60": dict(T=200, M=60, nbase=4, seed=82, localpen=None, scale=1.0),
    "t200m60_glocal": dict(T=200, M=60, nbase=4, seed=82, localpen=2.0, scale=1.0),
    "t500m1": dict(T=500, M=1, nbase=4, seed=83, localpen=1.0, scale=1.0),
    "t64m64": dict(T=64, M=64, nbase=4, seed=84, localpen=None, scale=1.0),     # every block steps
    "t40m70_too_long": dict(T=40, M=70, nbase=4, seed=85, localpen=None, scale=1.0),   # no global path
    "t900m300_nb2": dict(T=900, M=300, nbase=2, seed=86, localpen=0.7, scale=1.0),
    "t1500m700_ties": dict(T=1500, M=700, nbase=4, seed=87, localpen=3.0, scale=-1.0),  # integer scores
    "t3000m1300": dict(T=3000, M=1300, nbase=4, seed=88, localpen=4.0, scale=1.0),
    # junk at both ends of the signal: the glocal mapping clips it (path -1), the global one cannot
    "t400m90_clip": dict(T=400, M=90, nbase=4, seed=89, localpen=1.5, scale=1.0, junk=(0.25, 0.15)),
    "t400m90_clip_global": dict(T=400, M=90, nbase=4, seed=89, localpen=None, scale=1.0, junk=(0.25, 0.15)),
    "t2500m800_clip": dict(T=2500, M=800, nbase=4, seed=90, localpen=0.5, scale=1.0, junk=(0.1, 0.3)),
}


def remap_inputs(spec):
    """(scores (T, K) float32, bases (M,) int).  scale < 0: scores rounded to integers in
    [-2, 2] so that ties are everywhere (the strict '<' of the traceback decides)."""
    nb = spec["nbase"]
    sc = synth.scores(spec["T"], 1, 2 * nb * (nb + 1), spec["seed"])[:, 0, :]
    if spec["scale"] < 0:
        sc = np.round(sc * np.float32(0.4)).astype(np.float32)
    if spec.get("junk"):
        head, tail = (int(f * spec["T"]) for f in spec["junk"])
        sc[:head] -= np.float32(7.0)
        sc[spec["T"] - tail:] -= np.float32(7.0)
    bases = synth.randint(spec["seed"], 21, spec["M"], nb)
    rep = synth.uniform01(spec["seed"], 22, spec["M"]) < 0.3
    for p in range(1, spec["M"]):
        if rep[p]:
            bases[p] = bases[p - 1]
    return np.ascontiguousarray(sc), bases


def realnet_catmod_inputs(gold, tag="real", seed=97, logit_scale=0.2):
    """The trained network's canonical scores (tests/golden/realnet.npz) with the six modification columns of
    ACGTZY beside them: synthetic logits, per-base log-softmax -- what GlobalNormFlipFlopCatMod emits
    (layers.py:1627-1640) -- and modification categories drawn on the reads' true A / C positions.  The mod
    columns are regenerated from the seed; the fixture holds the genuine reference's OUTPUTS."""
    can = gold[tag + "/scores"]
    T, N, _ = can.shape
    lens = gold[tag + "/seqlens"].astype(np.int32)
    inp = dict(scores=np.concatenate([can, synth.scores(T, N, 6, seed)], axis=2),
               seqs=gold[tag + "/seqs"].astype(np.int64), seqlens=lens,
               mod_cats=synth.mod_cats(gold[tag + "/bases"].astype(np.int64), seed, NMODS),
               can_mods_offsets=synth.can_mods_offsets(NMODS), mod_cat_weights=np.full(6, 8.0, dtype=np.float32))
    return synth.normalise_mod_columns(inp, logit_scale=logit_scale)
