#!/usr/bin/env python
"""Numpy model of kernel A's LINEAR-DOMAIN band mode (csrc/crf_band.hip, round 3).

Same schedule as tests/helpers/crf_skew_model.py (chunks of PW cells, time blocks of KB steps, the
live band of (chunk, block) pairs, a boundary ring between neighbouring chunks) but the
arithmetic of c_crf_flipflop.c:43-78 / 150-182 / 372-413 is done on

    value(cell) = m * 2^e,   m float32, e int32 PER CELL, e fixed for the KB steps of a block

so a lattice step is  m' = m * es + m_upstream * (em * 2^(e_upstream - e))  -- two multiply-adds
per cell with no exp / log on the serial chain.  The score row is exponentiated ONCE per row
(es / em are gathers from exp2(c * row)); power-of-two rescaling is exact in fp32 and the
exponents add exactly in int32.

  * block start: every live cell is renormalised to m in [2^(BIAS-1), 2^BIAS) (frexp / ldexp, no
    wave reduction); a dead cell (m == 0: beyond the diagonal front, or killed by a zero weight)
    takes the exponent of the nearest live cell UPSTREAM of it (forward: lower positions), else
    the incoming boundary cell's exponent -- so the cells that come alive during the block start
    in the frame of the cell that feeds them;
  * what the gradient pass needs is ONE checkpoint column (m, e) per block and the KB boundary
    cells per (chunk, block): it recomputes the block's columns itself (KB - 1 cheap steps each
    way) instead of reading two stored lattices (round 2: 10.8x / 43x the algorithmic bytes);
  * the posterior of row t is  (F_t[p] es) B_{t+1}[p]  (stay at p)  and  (F_t[p-1] em) B_{t+1}[p]
    (move INTO p): both products of the forward step's own two terms with the same lane's backward
    cell, scaled by 2^(eF + eB - floor(log2 Z));
  * a read whose cells leave the fp32 range inside a block (a live cell found zero / denormal /
    non-finite at the next block start, or a non-finite score) is FLAGGED and redone by the
    log-domain checkpoint kernel (crf_kernels.hip): the linear path is exact or says so.

Development aid + CPU-checkable statement of the algorithm (tests/test_band_model.py); the
product path is the HIP kernel, which mirrors it name for name.
"""
import numpy as np

from tests.helpers.crf_skew_model import windows

f32 = np.float32
LOG2E = f32(1.4426950408889634)
BIAS = 0
DMIN, DMAX = -150, 126
TINY = np.finfo(np.float32).tiny


def ldexp32(m, k):
    with np.errstate(over="ignore", under="ignore"):
        return np.ldexp(m.astype(f32), np.asarray(k, dtype=np.int64)).astype(f32)


# Every step weight carries a factor 2^-WBIAS (folded into the exponential's argument: one fma).  It
# CENTRES the weights' range -- 2^(+-7.2) for |sharp score| <= 5 becomes 2^(-7.2 - WBIAS) .. 2^(7.2 - WBIAS)
# -- so that growth (1 + 2^KLIP) 2^(7.2 - WBIAS) per step and decay 2^-(7.2 + WBIAS) per step use both halves
# of fp32's exponent range: 12-step blocks fit at WBIAS = 3 (12 x 10.2 = 123 bits up, 12 x 10.2 down) where
# 8 steps was the most without it.  All lattice values at time t are scaled by 2^(-WBIAS t): posteriors
# (ratios to Z) do not see it, the two scores get WBIAS T added back.
WBIAS = 0


def exp_rows(scores, c):
    """(T, S+2): exp2(c * scores - WBIAS), column S = 0 (dead transition), column S+1 = 1."""
    T, S = scores.shape
    out = np.zeros((T, S + 2), dtype=f32)
    with np.errstate(over="ignore", under="ignore"):
        out[:, :S] = np.exp2((scores.astype(f32) * f32(c) - f32(WBIAS)).astype(f32)).astype(f32)
    out[:, S + 1] = 1
    return out


class Read:
    """Per-position transition ids of one read, padded to whole chunks."""

    def __init__(self, stay, move, L, PW, S, mod=None, modfact=None):
        self.L, self.PW, self.S = L, PW, S
        self.W = (L + PW - 1) // PW
        P = self.W * PW
        self.st = np.full(P, S)
        self.st[:L] = stay[:L]
        self.mvout = np.full(P, S)
        self.mvout[:L - 1] = move[:L - 1]
        self.mvin = np.full(P, S)
        self.mvin[1:L] = move[:L - 1]
        self.has_mod = mod is not None
        if self.has_mod:
            self.mdout = np.full(P, S + 1)
            self.mdout[:L - 1] = mod[:L - 1]
            self.mdin = np.full(P, S + 1)
            self.mdin[1:L] = mod[:L - 1]
            self.fwout = np.zeros(P, dtype=f32)
            self.fwout[:L - 1] = modfact[:L - 1]
            self.fwin = np.zeros(P, dtype=f32)
            self.fwin[1:L] = modfact[:L - 1]


def weights(rd, erow, raw, t, sl, forward, c_can, c_mod):
    """Stay and move weights of the cells `sl` for row t (move INTO the cell forward, OUT of it backward)."""
    es = erow[t][rd.st[sl]]
    mv = rd.mvin[sl] if forward else rd.mvout[sl]
    if not rd.has_mod:
        return es, erow[t][mv]
    md = rd.mdin[sl] if forward else rd.mdout[sl]
    fw = (rd.fwin[sl] if forward else rd.fwout[sl]) * f32(c_mod)
    rawS = np.concatenate([raw[t], [f32(-1e30), f32(0)]]).astype(f32)
    with np.errstate(over="ignore", under="ignore"):
        em = np.exp2((rawS[mv] * f32(c_can) + rawS[md] * fw - f32(WBIAS)).astype(f32)).astype(f32)
    em[mv == rd.S] = 0
    return es, em


KLIP = 6
LEM_MIN = -100       # cat-mod: the frame slope follows move weights down to 2^-100


def block_frame(m, e, forward, e_b, lem=None):
    """Block-start frames.  Own exponent eo = e + exponent(m) of every live cell; the frame is the
    K-Lipschitz envelope  f[p] = max(eo[p], f[upstream] - KLIP)  (a decayed prefix maximum along the
    flow; the boundary cell starts from the neighbouring chunk's frame e_b).  Then no cell can
    receive more than 2^KLIP times its frame unit per step: growth inside a block is bounded by
    the weights alone, whatever cliffs the lattice has.  A cell far below its upstream neighbours
    keeps a small mantissa (flushed to zero beyond 2^-126: it is about to be overwritten by their
    inflow).  Returns (m, f, d, nflush): d[c] = f[upstream of c] - f[c]  (<= KLIP)."""
    n = len(m)
    live = (m > 0) & np.isfinite(m)
    with np.errstate(all="ignore"):
        ex = np.frexp(m)[1].astype(np.int64)
    BIG = 1 << 40
    eo = np.where(live, e + ex - BIAS, -BIG)
    f = np.zeros(n, dtype=np.int64)
    cur = e_b if e_b is not None else -BIG
    for c in (range(n) if forward else range(n - 1, -1, -1)):
        cur = max(eo[c], cur - KLIP + (0 if lem is None else int(lem[c])))
        f[c] = cur
    f = np.where(f < -(BIG >> 1), 0, f)                     # nothing upstream and dead: any frame does
    mm = ldexp32(m, np.clip(e - f, -300, 300))
    nflush = int((live & (mm < TINY)).sum())
    if forward:
        up = np.concatenate([[e_b if e_b is not None else f[0]], f[:-1]])
    else:
        up = np.concatenate([f[1:], [e_b if e_b is not None else f[-1]]])
    d = np.clip(up - f, DMIN, DMAX)
    return mm, f, d, nflush


def step(m, sc, es, em, b_in, forward):
    mterm = (em * sc).astype(f32)
    up = np.concatenate([[b_in], m[:-1]]) if forward else np.concatenate([m[1:], [b_in]])
    with np.errstate(all="ignore"):
        return (m * es + up.astype(f32) * mterm).astype(f32)


def run_block(rd, erow, raw, T, KB, NORM, j, w, m, e, forward, pl, ring_m, ring_e, c_can, c_mod,
              waslive=None):
    """The KB steps of time block j on chunk w, from the cells (m, e) the chunk holds when the block
    starts (the sweep's running cells, or a checkpoint column: renormalising a normalised column
    changes nothing).  ring_m[i] / ring_e[sub]: the neighbouring chunk's boundary cell before step
    i and its exponent in sub-block `sub` (only read when pl).
    Returns (cols, out_m, out_e, m, e, bad): cols[i] = (m, e, sc) of the column step i consumes,
    out_m[i] / out_e[sub] what this chunk hands to ITS neighbour."""
    sl = slice(w * rd.PW, (w + 1) * rd.PW)
    nvalid = min(KB, T - j * KB)
    cols = [None] * nvalid
    out_m = np.zeros(KB, dtype=f32)
    out_e = np.zeros(KB // NORM, dtype=np.int64)
    bad = False
    sc = None
    for i in (range(nvalid) if forward else range(nvalid - 1, -1, -1)):
        sub = i // NORM
        first = (i % NORM == 0) if forward else (i % NORM == NORM - 1 or i == nvalid - 1)
        if first:
            bad |= bool((~np.isfinite(m)).any())
            lem = None
            if rd.has_mod:
                steps = [k for k in range(nvalid) if k // NORM == sub]
                emx = np.max([weights(rd, erow, raw, j * KB + k, sl, forward, c_can, c_mod)[1] for k in steps], axis=0)
                with np.errstate(all="ignore"):
                    lem = np.where(emx > 0, np.clip(np.frexp(emx)[1], LEM_MIN, 0), 0)
            m, e, d, nf = block_frame(m, e, forward, int(ring_e[sub]) if pl else None, lem)
            if waslive is not None:
                waslive[0] += nf
            sc = ldexp32(np.ones(rd.PW, dtype=f32), d)
            out_e[sub] = e[-1] if forward else e[0]
        t = j * KB + i
        cols[i] = (m, e, sc)
        out_m[i] = m[-1] if forward else m[0]
        es, em = weights(rd, erow, raw, t, sl, forward, c_can, c_mod)
        m = step(m, sc, es, em, ring_m[i] if pl else f32(0), forward)
    return cols, out_m, out_e, m, e, bad


def sweep(rd, erow, raw, T, KB, NORM, forward, c_can=1.0, c_mod=1.0):
    L, PW, W = rd.L, rd.PW, rd.W
    NB = (T + KB - 1) // KB
    win = windows(L, T, PW, KB)
    P = W * PW
    M = np.zeros(P, dtype=f32)
    M[0 if forward else L - 1] = 1
    E = np.zeros(P, dtype=np.int64)
    waslive = np.zeros(1, dtype=np.int64)                   # (count of flushed live cells: diagnostics)
    ck_m = np.full((NB, P), np.nan, dtype=f32)
    ck_e = np.zeros((NB, P), dtype=np.int64)
    bnd_m = np.zeros((NB, W, KB), dtype=f32)
    bnd_e = np.zeros((NB, W, KB // NORM), dtype=np.int64)
    stored = np.zeros((NB, W), dtype=bool)
    bad = False
    for j in (range(NB) if forward else range(NB - 1, -1, -1)):
        for w in (range(W) if forward else range(W - 1, -1, -1)):
            if not (win[w][0] <= j <= win[w][1]):
                continue
            stored[j, w] = True
            src = w - 1 if forward else w + 1
            pl = 0 <= src < W and win[src][0] <= j <= win[src][1]
            sl = slice(w * PW, (w + 1) * PW)
            srcc = min(max(src, 0), W - 1)
            cols, bnd_m[j, w], bnd_e[j, w], M[sl], E[sl], b = run_block(
                rd, erow, raw, T, KB, NORM, j, w, M[sl], E[sl], forward, pl, bnd_m[j, srcc], bnd_e[j, srcc],
                c_can, c_mod, waslive)
            bad |= b
            nvalid = min(KB, T - j * KB)
            ck_m[j, sl], ck_e[j, sl] = cols[0 if forward else nvalid - 1][:2]
    bad |= bool((~np.isfinite(M)).any())
    pend = L - 1 if forward else 0
    with np.errstate(all="ignore"):
        score = float(E[pend]) + float(np.log2(np.float64(M[pend])))
    bad |= not np.isfinite(score)
    return dict(ck_m=ck_m, ck_e=ck_e, bnd_m=bnd_m, bnd_e=bnd_e, stored=stored, score=score, bad=bad, nflush=int(waslive[0]))


def posterior(rd, erow, raw, T, KB, NORM, F, B, c_can=1.0, c_mod=1.0):
    """Normalised posterior rows (T, S) by recomputing every block from its two checkpoint
    columns and the boundary cells."""
    L, PW, W, S = rd.L, rd.PW, rd.W, rd.S
    NB = (T + KB - 1) // KB
    win = windows(L, T, PW, KB)
    zexp = int(np.floor(F["score"]))
    out = np.zeros((T, S + 2), dtype=f32)
    total = np.zeros(T, dtype=f32)
    for j in range(NB):
        nvalid = min(KB, T - j * KB)
        for w in range(W):
            if not (F["stored"][j, w] and B["stored"][j, w]):
                continue
            sl = slice(w * PW, (w + 1) * PW)
            plF = w > 0 and win[w - 1][0] <= j <= win[w - 1][1]
            plB = w + 1 < W and win[w + 1][0] <= j <= win[w + 1][1]
            colsF = run_block(rd, erow, raw, T, KB, NORM, j, w, F["ck_m"][j, sl], F["ck_e"][j, sl], True, plF,
                              F["bnd_m"][j, max(w - 1, 0)], F["bnd_e"][j, max(w - 1, 0)], c_can, c_mod)[0]
            colsB = run_block(rd, erow, raw, T, KB, NORM, j, w, B["ck_m"][j, sl], B["ck_e"][j, sl], False, plB,
                              B["bnd_m"][j, min(w + 1, W - 1)], B["bnd_e"][j, min(w + 1, W - 1)], c_can, c_mod)[0]
            for i in range(nvalid):
                t = j * KB + i
                fv, eF, scF = colsF[i]              # column t
                bv, eB, _ = colsB[i]                # column t + 1 (what backward step i consumes)
                es, em = weights(rd, erow, raw, t, sl, True, c_can, c_mod)
                b_in = F["bnd_m"][j, w - 1, i] if plF else f32(0)
                up = np.concatenate([[b_in], fv[:-1]]).astype(f32)
                with np.errstate(all="ignore"):
                    Fs = (fv * es).astype(f32)
                    Fm = (up * (em * scF).astype(f32)).astype(f32)
                    bs = ldexp32(bv, eF + eB - zexp)
                    ps, pm = (Fs * bs).astype(f32), (Fm * bs).astype(f32)
                np.add.at(out[t], rd.st[sl], ps)
                np.add.at(out[t], rd.mvin[sl], pm)
                total[t] += ps.sum(dtype=f32) + pm.sum(dtype=f32)
                if rd.has_mod:
                    np.add.at(out[t], rd.mdin[sl], (pm * rd.fwin[sl]).astype(f32))
    with np.errstate(all="ignore"):
        return out[:, :S] / total[:, None], np.log2(total.astype(np.float64)) + zexp


def crf_model(scores, stay, move, L, PW=4, KB=8, NORM=8, sharp=1.0, mod=None, modfact=None, sharp_mod=1.0):
    """(cost, grad (T, S), info) of one read; info['bad'] = the linear path flags this read."""
    T, S = scores.shape
    c_can, c_mod = f32(sharp) * LOG2E, f32(sharp_mod) * LOG2E
    erow = exp_rows(scores, c_can)
    rd = Read(stay, move, L, PW, S, mod, modfact)
    raw = scores.astype(f32)
    F = sweep(rd, erow, raw, T, KB, NORM, True, c_can, c_mod)
    B = sweep(rd, erow, raw, T, KB, NORM, False, c_can, c_mod)
    bad = F["bad"] or B["bad"]
    info = dict(bad=bad, scoreF=F["score"], scoreB=B["score"])
    if bad:
        return None, None, info
    post, rowz = posterior(rd, erow, raw, T, KB, NORM, F, B, c_can, c_mod)
    info["rowz_dev"] = float(np.abs(rowz - F["score"]).max())
    info["nflush"] = F["nflush"] + B["nflush"]
    score2 = 0.5 * (F["score"] + B["score"]) + float(WBIAS) * T
    cost = -(score2 * np.log(2.0)) / T / sharp
    return cost, -post / T, info
