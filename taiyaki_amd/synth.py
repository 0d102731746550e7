"""Deterministic synthetic inputs for the flip-flop hot path (host side, numpy).

Counter-based (splitmix64) so that every test, the golden-fixture script and
``bench.py`` regenerate bit-identical inputs from ``(seed, shape)`` -- large
tensors are never committed.  The distributions mirror the reference's own
``SPEED_TEST`` harness (taiyaki/ctc/c_crf_flipflop.c:802-833): scores iid
U(-5, 5); ``seqlen_i = floor(T (1 + (i - N/2)/(5N)) / 2)``; bases iid uniform,
flip-flop coded (taiyaki/flipflopfings.py:56-78).
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """Vectorised splitmix64 finaliser of (x + golden) on uint64 arrays."""
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream_base(seed, stream):
    with np.errstate(over="ignore"):
        s = splitmix64(np.array([np.uint64(seed)], dtype=np.uint64))
        s = splitmix64(s + np.uint64(stream) * np.uint64(0xD1342543DE82EF95))
    return s[0]


def uniform01(seed, stream, n, offset=0):
    """n float32 values in [0, 1) (24 random bits each) from counters offset..offset+n."""
    base = _stream_base(seed, stream)
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        with np.errstate(over="ignore"):
            ctr = np.arange(lo + offset, hi + offset, dtype=np.uint64) + base
        z = splitmix64(ctr)
        out[lo:hi] = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return out


def randint(seed, stream, n, high):
    base = _stream_base(seed, stream)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + base
    return (splitmix64(ctr) >> np.uint64(33)).astype(np.int64) % int(high)


def scores(T, N, S, seed, lo=-5.0, hi=5.0):
    """(T, N, S) float32 iid U(lo, hi) -- SPEED_TEST logprob (c_crf_flipflop.c:806-810)."""
    u = uniform01(seed, 1, T * N * S)
    return (np.float32(hi - lo) * u + np.float32(lo)).reshape(T, N, S)


def flipflop_code(bases, nbase=4):
    """flipflopfings.py:56-78 restated: +nbase at even positions within runs."""
    bases = np.asarray(bases, dtype=np.int64)
    out = bases.copy()
    run = 0
    for p in range(len(bases)):
        run = run + 1 if (p > 0 and bases[p] == bases[p - 1]) else 0
        if run & 1:
            out[p] += nbase
    return out


def speedtest_seqlens(T, N):
    """c_crf_flipflop.c:813 -- 0.45..0.55 T."""
    i = np.arange(N, dtype=np.float32)
    return (np.float32(T) * (1 + (i - np.float32(0.5) * N) / (np.float32(5.0) * N)) / 2
            ).astype(np.int32)


def realistic_seqlens(T, N, seed, chunk_len, samples_per_base=9.0):
    """L_i ~ U(0.8, 1.2) * chunk_len / samples_per_base, clipped to L < T/1.1
    (the reference's own chunk filter, signal_mapping.py:699-703)."""
    u = uniform01(seed, 5, N)
    L = ((0.8 + 0.4 * u) * chunk_len / samples_per_base).astype(np.int32)
    return np.clip(L, 1, int(T / 1.1) - 1).astype(np.int32)


def sequences(seqlens, seed, nbase=4):
    """Concatenated flip-flop coded sequences for the given lengths.

    Returns (seqs (sum L,) int64, bases (sum L,) int64).
    """
    seqlens = np.asarray(seqlens, dtype=np.int64)
    total = int(seqlens.sum())
    bases = randint(seed, 2, total, nbase)
    codes = np.empty(total, dtype=np.int64)
    off = 0
    for L in seqlens:
        codes[off:off + L] = flipflop_code(bases[off:off + L], nbase)
        off += L
    return codes, bases


def mod_cats(bases, seed, nmods_per_base=(1, 1, 0, 0), p=0.3):
    """Bernoulli(p) modified-base category on bases that have a modification
    (ACGTZY: 6mA on A, 5mC on C).  0 = canonical, k = k-th mod of that base
    (taiyaki/layers.py:1441-1460)."""
    bases = np.asarray(bases, dtype=np.int64)
    u = uniform01(seed, 3, len(bases))
    nm = np.asarray(nmods_per_base, dtype=np.int64)[bases % len(nmods_per_base)]
    k = 1 + (uniform01(seed, 4, len(bases)) * np.maximum(nm, 1)).astype(np.int64)
    return np.where((nm > 0) & (u < p), np.minimum(k, np.maximum(nm, 1)), 0).astype(np.int64)


def can_mods_offsets(nmods_per_base=(1, 1, 0, 0)):
    """layers.py:1495-1497: [0,2,4,5,6] for ACGTZY."""
    return np.concatenate([[0], np.cumsum(1 + np.asarray(nmods_per_base))]).astype(np.int32)


def signal_chunks(chunk_len, N, seed):
    """(chunk_len, N, 1) float32 ~ N(0,1): standardised signal (docs/abinitio.rst:106-118).
    Box-Muller on the counter stream."""
    n = chunk_len * N
    u1 = uniform01(seed, 6, n)
    u2 = uniform01(seed, 7, n)
    r = np.sqrt(-2.0 * np.log(np.maximum(u1, np.float32(2.0 ** -24))))
    return (r * np.cos(2 * np.pi * u2)).astype(np.float32).reshape(chunk_len, N, 1)


def crf_case(T, N, seed, nbase=4, nmods_per_base=None, seqlens=None):
    """One complete synthetic operator input set.

    Returns dict(scores, seqs, seqlens[, mod_cats, can_mods_offsets, mod_cat_weights]).
    """
    S = 2 * nbase * (nbase + 1)
    if nmods_per_base is not None:
        S += nbase + int(np.sum(nmods_per_base))
    if seqlens is None:
        seqlens = speedtest_seqlens(T, N)
    seqlens = np.asarray(seqlens, dtype=np.int32)
    seqs, bases = sequences(seqlens, seed, nbase)
    out = dict(scores=scores(T, N, S, seed), seqs=seqs, seqlens=seqlens)
    if nmods_per_base is not None:
        out["mod_cats"] = mod_cats(bases, seed, nmods_per_base)
        out["can_mods_offsets"] = can_mods_offsets(nmods_per_base)
        out["mod_cat_weights"] = np.full(nbase + int(np.sum(nmods_per_base)), 8.0,
                                         dtype=np.float32)
    return out


def confident_scores(inp, seed, on=4.0, off=-3.0, noise=1.0, nbase=4, bursty=False, move_times=None):
    """Scores of a TRAINED network, not of a freshly initialised one: every read follows one alignment
    (its L - 1 moves at random block positions, `bursty`: in runs, as a strand that speeds up and
    stalls; `move_times`: a callable n -> the sorted blocks at which read n moves, for alignments built on
    purpose), the transition the alignment takes at a block scores `on` (+- noise), every other one
    `off` (+- noise) -- the 5 tanh range used to its ends.  Replaces inp["scores"] in place."""
    T, N, S = inp["scores"].shape
    rng = np.random.RandomState(seed)
    sc = (off + noise * rng.uniform(-1, 1, size=(T, N, S))).astype(np.float32)
    ns = 2 * nbase
    off_seq = np.concatenate([[0], np.cumsum(inp["seqlens"])])
    for n in range(N):
        L = int(inp["seqlens"][n])
        if L == 0 or L - 1 > T:
            continue
        codes = inp["seqs"][off_seq[n]:off_seq[n] + L].astype(int)
        if move_times is not None:
            moves = np.asarray(move_times(n), dtype=int)
            assert len(moves) == L - 1 and len(set(moves.tolist())) == L - 1 and moves.min() >= 0 and moves.max() < T
        elif bursty:
            w = np.repeat(rng.uniform(0.05, 1.0, size=T // 40 + 1) ** 3, 40)[:T]
            moves = np.sort(rng.choice(T, size=L - 1, replace=False, p=w / w.sum()))
        else:
            moves = np.sort(rng.choice(T, size=L - 1, replace=False))
        is_move = np.zeros(T, dtype=bool)
        is_move[moves] = True
        p = 0
        for t in range(T):
            c = codes[p]
            if is_move[t]:
                tid = c + min(codes[p + 1], nbase) * ns            # flipflopfings.py:6-17
                p += 1
            else:
                tid = c + min(c, nbase) * ns                       # flipflopfings.py:20-31
            sc[t, n, tid] = on + noise * rng.uniform(-1, 1)
    inp["scores"][:, :, :S] = sc
    return inp


def normalise_mod_columns(inp, ncan=40, logit_scale=0.2):
    """Turn the free scores of a cat-mod case's mod columns into what GlobalNormFlipFlopCatMod emits
    there (layers.py:1627-1640): per canonical base a log-softmax over {unmodified, its
    modifications}.  The free scores (U(-5, 5)) times `logit_scale` are the logits: 0.2 gives the
    +-1 a freshly initialised layer produces (log-probabilities around log 1/2); 1.0 makes every row
    disagree violently with its neighbours about every modification (log-probabilities down to -10,
    times the reference's initial mod_factor of 8) -- a stress case, not a network.  In place;
    returns inp."""
    offs = np.asarray(inp["can_mods_offsets"])
    sc = inp["scores"]
    for b in range(len(offs) - 1):
        blk = sc[:, :, ncan + offs[b]:ncan + offs[b + 1]].astype(np.float64) * logit_scale
        blk -= np.log(np.exp(blk).sum(axis=2, keepdims=True))
        sc[:, :, ncan + offs[b]:ncan + offs[b + 1]] = blk.astype(np.float32)
    return inp


def mapped_reads(nreads, seed, nlabel=4, mean_reflen=400, mean_dwell=9, clip_prob=0.5,
                 slip_prob=0.03, long_dwell_prob=0.01):
    """Synthetic per-read dictionaries with the fields of the reference's mapped-signal
    format (signal_mapping.py:26-33, docs/FILE_FORMATS.md:43-75): int16 Dacs, int32
    Ref_to_signal (reflen + 1, non-decreasing; -1 = reference start not mapped, siglen + 1 =
    reference end not mapped), int16 Reference, and the five scaling floats.  Dwells are
    1 + geometric-like around `mean_dwell`, with a few zero-dwell ("slip") bases and a few very
    long ones so that every chunk filter has something to reject; reads of very different
    lengths so that some are too short for a chunk."""
    reads = []
    for r in range(nreads):
        s = seed * 1000003 + r
        reflen = int(8 + randint(s, 11, 1, 2 * mean_reflen)[0] * (0.05 if r % 7 == 3 else 1.0))
        u = uniform01(s, 12, reflen)
        dw = (1 + np.floor(-np.log(np.maximum(u, 1e-6)) * (mean_dwell - 1))).astype(np.int64)
        kind = uniform01(s, 13, reflen)
        dw[kind < slip_prob] = 0
        dw[kind > 1.0 - long_dwell_prob] *= 12
        head_ref, tail_ref = 0, 0
        cu = uniform01(s, 14, 4)
        if cu[0] < clip_prob:
            head_ref = int(cu[1] * min(20, reflen // 4))
        if cu[2] < clip_prob:
            tail_ref = int(cu[3] * min(20, reflen // 4))
        pad = randint(s, 15, 2, 50)
        mapped_dw = dw[head_ref:reflen - tail_ref]
        sig0 = int(pad[0])
        pos = sig0 + np.concatenate([[0], np.cumsum(mapped_dw)])
        siglen = int(pos[-1] + pad[1])
        rts = np.empty(reflen + 1, dtype=np.int32)
        rts[:head_ref] = -1
        rts[head_ref:head_ref + len(pos)] = pos
        rts[head_ref + len(pos):] = siglen + 1
        lab = randint(s, 16, reflen, nlabel)
        rep = uniform01(s, 17, reflen) < 0.3            # homopolymer runs exercise the flop states
        for p in range(1, reflen):
            if rep[p]:
                lab[p] = lab[p - 1]
        fl = uniform01(s, 18, 5)
        reads.append(dict(
            read_id="synth-%d-%d" % (seed, r),
            Dacs=(randint(s, 19, siglen, 1200) + 200).astype(np.int16),
            Ref_to_signal=rts, Reference=lab.astype(np.int16),
            offset=float(np.float32(-20 + 40 * fl[0])), range=float(np.float32(1000 + 800 * fl[1])),
            digitisation=8192.0, shift_frompA=float(np.float32(60 + 60 * fl[2])),
            scale_frompA=float(np.float32(8 + 10 * fl[3]))))
    return reads
