#!/usr/bin/env python
"""Numpy model of kernel A's schedule (csrc/crf_kernels.hip, "banded skewed sweep"):

  * positions are cut into chunks of PW cells (one wavefront each on the GPU);
  * chunk w runs time block j (KB steps) in phase j + w (forward) resp. phase
    (NB - 1 - j) + (W - 1 - w) (backward): ONE workgroup barrier per KB steps; the
    boundary cell travels through a two-slot edge ring written one phase earlier;
  * every chunk keeps its own INTEGER log2 offset, renormalised every NORM steps by
    floor(max(own cells, incoming edge cells)) -- exact in fp32, no fp64 anywhere;
  * only the band of (chunk, block) pairs that lie on some complete path
    (p <= t  and  L - 1 - p <= T - t) is computed or stored;
  * the posterior pass treats (row, chunk) pairs independently: the 2*PW (3*PW for
    cat-mod) transition instances of a chunk are evaluated in an order SORTED by
    transition id, so the per-id sums are differences of one prefix scan.

This file is a development aid and a CPU-checkable statement of the algorithm
(tests/test_band_model.py runs it against the oracle); the product path is the HIP
kernel, which it mirrors name for name.
"""
import numpy as np

f32 = np.float32
NEG = f32(-1e30) * f32(1.4426950408889634)
LOG2E = f32(1.4426950408889634)


def lse2(a, b):
    mx = np.maximum(a, b)
    d = -np.abs(a - b)
    return (mx + np.log2(f32(1.0) + np.exp2(d).astype(f32)).astype(f32)).astype(f32)


def windows(L, T, PW, KB):
    """Per chunk: first / last live time block (inclusive); j0 > j1 = never live."""
    W = (L + PW - 1) // PW
    NB = (T + KB - 1) // KB
    out = []
    for w in range(W):
        a, b = w * PW, min(w * PW + PW - 1, L - 1)
        if L > T + 1:                       # no complete path: nothing to trim by
            out.append((0, NB - 1))
            continue
        tlo, thi = max(0, a - 1), min(T - 1, b + T - L + 1)
        out.append((tlo // KB, thi // KB) if tlo <= thi else (1, 0))
    return out


def sweep(lp2, stay, move, L, PW, KB=8, NORM=4, forward=True):
    """lp2: (T, S+2) float32 scores * c with sentinel columns S (NEG) and S+1 (0).
    Returns (lat[T][W*PW] stored columns, off[T//NORM + 1][W] ints, stored mask, score2)."""
    T = lp2.shape[0]
    W = (L + PW - 1) // PW
    NB = (T + KB - 1) // KB
    S = lp2.shape[1] - 2
    win = windows(L, T, PW, KB)
    P = W * PW
    st = np.full(P, S)
    st[:L] = stay[:L]
    mv = np.full(P, S)
    mv[:L - 1] = move[:L - 1]
    cell = np.full(P, NEG, dtype=f32)
    cell[0 if forward else L - 1] = 0
    off = np.zeros(W, dtype=np.int64)
    lat = np.full((T, P), np.nan, dtype=f32)
    offs = np.zeros(((T + NORM - 1) // NORM, W), dtype=np.int64)
    stored = np.zeros((NB, W), dtype=bool)
    E = np.full((W, 2, KB), NEG, dtype=f32)
    Eoff = np.zeros((W, 2, KB // NORM), dtype=np.int64)
    for ph in range(NB + W - 1):
        Enew, Eoffnew = E.copy(), Eoff.copy()       # writes become visible at the barrier
        for w in range(W):
            j = ph - w if forward else (NB - 1) - (ph - (W - 1 - w))
            if not (win[w][0] <= j <= win[w][1]):
                continue
            stored[j, w] = True
            src = w - 1 if forward else w + 1
            pl = 0 <= src < W and win[src][0] <= j <= win[src][1]
            sl = slice(w * PW, (w + 1) * PW)
            steps = range(KB) if forward else range(KB - 1, -1, -1)
            for i in steps:
                t = j * KB + i
                if t >= T:
                    continue
                sub = i // NORM
                first = (i % NORM == 0) if forward else (i % NORM == NORM - 1 or t == T - 1)
                if first:
                    # renormalise: own cells and the incoming edge cells of this sub-block
                    mx = cell[sl].max()
                    if pl:
                        lo = sub * NORM
                        ein = E[src, j & 1, lo:min(lo + NORM, KB)]
                        ein = ein[[q for q in range(len(ein)) if j * KB + lo + q < T]]
                        delta = f32(Eoff[src, j & 1, sub] - off[w])
                        mx = max(mx, (ein + delta).astype(f32).max())
                    m = int(np.floor(mx)) if mx > -1e29 else 0
                    cell[sl] = (cell[sl] - f32(m)).astype(f32)
                    off[w] += m
                    offs[t // NORM, w] = off[w]
                    Eoffnew[w, j & 1, sub] = off[w]
                lat[t, sl] = cell[sl]
                row = lp2[t]
                c = cell[sl]
                if forward:
                    Enew[w, j & 1, i] = c[-1]
                    edge = NEG
                    if pl:
                        edge = f32(E[src, j & 1, i] + f32(Eoff[src, j & 1, sub] - off[w]))
                    left = np.concatenate([[edge], c[:-1]]).astype(f32)
                    mvin = np.concatenate([[mv[w * PW - 1] if w > 0 else S], mv[sl][:-1]])
                    cell[sl] = lse2((row[st[sl]] + c).astype(f32), (row[mvin] + left).astype(f32))
                else:
                    Enew[w, j & 1, i] = c[0]
                    edge = NEG
                    if pl:
                        edge = f32(E[src, j & 1, i] + f32(Eoff[src, j & 1, sub] - off[w]))
                    right = np.concatenate([c[1:], [edge]]).astype(f32)
                    cell[sl] = lse2((row[st[sl]] + c).astype(f32), (row[mv[sl]] + right).astype(f32))
        E, Eoff = Enew, Eoffnew
    if forward:
        score = off[(L - 1) // PW] + float(cell[L - 1])
    else:
        score = off[0] + float(cell[0])
    return lat, offs, stored, score


def chunk_records(stay, move, L, w, PW, S):
    """Sorted transition instances of chunk w: (idxF, idxB, id), segment ends per id."""
    a = w * PW
    inst = []
    for p in range(a, a + PW):
        inst.append((stay[p] if p < L else 63, p - a, p - a))            # stay
    for p in range(a, a + PW):
        inst.append((move[p] if p < L - 1 else 63, p - a, p - a + 1))    # move
    order = sorted(range(len(inst)), key=lambda e: (inst[e][0], e))
    rec = [inst[e] for e in order]
    keys = np.array([r[0] for r in rec])
    segend = np.array([int((keys <= s).sum()) for s in range(64)])
    return rec, segend


def posterior(lp2, stay, move, L, PW, F, offF, B, offB, scoreF, KB=8, NORM=4):
    """Gradient rows (T, S) of -score/T ... here: normalised posteriors (rows sum to 1)."""
    T = lp2.shape[0]
    S = lp2.shape[1] - 2
    W = (L + PW - 1) // PW
    recs = [chunk_records(stay, move, L, w, PW, S) for w in range(W)]
    out = np.zeros((T, S), dtype=f32)
    trim = L <= T + 1
    for t in range(T):
        acc = np.zeros(64, dtype=f32)
        total = f32(0)
        for w in range(W):
            a, b = w * PW, min(w * PW + PW - 1, L - 1)
            if trim and not (a <= t and b + 1 >= L - T + t):
                continue
            ct = f32(scoreF - offF[t // NORM, w] - offB[t // NORM, w])
            sF = (F[t, a:a + PW] - ct).astype(f32)
            sB = np.full(PW + 1, NEG, dtype=f32)
            sB[:PW] = B[t, a:a + PW]
            right_live = w + 1 < W and (not trim or t >= a + PW - 1)
            if right_live:
                sB[PW] = f32(B[t, a + PW] + f32(offB[t // NORM, w + 1] - offB[t // NORM, w]))
            rec, segend = recs[w]
            ids = np.array([min(r[0], S) for r in rec])
            v = np.exp2((sF[[r[1] for r in rec]] + sB[[r[2] for r in rec]] + lp2[t][ids]).astype(f32)).astype(f32)
            pref = np.cumsum(v, dtype=f32)
            Pk = np.array([pref[e - 1] if e > 0 else f32(0) for e in segend], dtype=f32)
            col = Pk - np.concatenate([[f32(0)], Pk[:-1]])
            acc += col.astype(f32)
            total += Pk[S - 1]
        out[t] = acc[:S] / total
    return out


def crf_model(scores, stay, move, L, PW=4, KB=8, NORM=4, sharp=1.0):
    T, S = scores.shape
    lp2 = np.zeros((T, S + 2), dtype=f32)
    lp2[:, :S] = scores * (f32(sharp) * LOG2E)
    lp2[:, S] = NEG
    F, offF, stF, scoreF = sweep(lp2, stay, move, L, PW, KB, NORM, True)
    B, offB, stB, scoreB = sweep(lp2, stay, move, L, PW, KB, NORM, False)
    post = posterior(lp2, stay, move, L, PW, F, offF, B, offB, scoreF, KB, NORM)
    score2 = 0.5 * (scoreF + scoreB)
    cost = -(score2 * np.log(2.0)) / T / sharp
    return cost, -post / T, (scoreF, scoreB, stF, stB)


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import oracle
    from taiyaki_amd import synth
    oracle.build()
    worst = 0.0
    for T, Ls, PW in ((20, [9, 1, 21, 20, 2], 4), (37, [12, 30, 38, 5], 4), (64, [33, 50, 7], 8),
                      (50, [25, 26], 64), (19, [20, 3], 2)):
        inp = synth.crf_case(T, len(Ls), 3 + T, seqlens=np.array(Ls, dtype=np.int32))
        oloss, ograd = oracle.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0)
        mv, stv = oracle.flipflop_indices(inp["seqs"], inp["seqlens"], 4)
        off = np.concatenate([[0], np.cumsum(Ls)])
        for n, L in enumerate(Ls):
            st = stv[off[n]:off[n] + L].astype(int)
            mo = mv[off[n] - n:off[n] - n + L - 1].astype(int)
            cost, grad, dbg = crf_model(inp["scores"][:, n], st, mo, L, PW)
            e1 = abs(cost - oloss[n]) / abs(oloss[n])
            e2 = np.abs(grad - ograd[:, n]).max()
            worst = max(worst, e1, e2)
            print("T=%d L=%d PW=%d  cost %.6f vs %.6f  rel %.2e  grad abs %.2e  (F %.4f B %.4f)"
                  % (T, L, PW, cost, oloss[n], e1, e2, dbg[0], dbg[1]))
    print("worst", worst)
