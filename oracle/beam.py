"""oracle.beam -- CPU restatement of the reference's hash beam search.  TEST INFRASTRUCTURE ONLY.

Restates `flipflop_beamsearch` (taiyaki/decodeutil/c_hashdecode.c:346-507), the guiding
backward pass `flipflop_backward` (c_flipflopfwdbwd.c:55-91) and the wrapper
`decodeutil.beamsearch` (decodeutil.pyx:9-51) in plain Python + numpy float32, pure loops (small
cases only).  Pinned against the GENUINE reference C compiled into oracle/_ref/libref_decodeutil.so
(tests/test_beamsearch.py): identical sequences and bit-identical scores, exact ties included.

The order of records with EQUAL scores decides which of them stay in the beam, and the reference
leaves it to its quicksort (taiyaki/decodeutil/qsort.h: median of second/middle/last, Sedgewick
partition, insertion sort below 16 elements, smaller subfile first).  Exact ties happen in about
one block in a hundred, so `qsort_order` below restates that procedure step by step and the two
sorts of a block (by hash, c_hashdecode.c:138-140; by score, :156-158) go through it -- the
restatement follows the reference through ties bit for bit.
"""
import ctypes
import os

import numpy as np

f32 = np.float32
_M = 0x880355F21E6D1965
_MASK = (1 << 64) - 1


def _mix(h):
    h ^= h >> 23
    h = (h * 0x2127599BF4325C37) & _MASK
    h ^= h >> 47
    return h


def chainfasthash64(h, val):
    """fasthash.c:95-103"""
    h ^= _mix(val & _MASK)
    h = (h * _M) & _MASK
    return _mix(h)


# How expf / log1pf are evaluated.  "libm": the host C library's own float functions, the ones the
# reference calls -- with them the restatement reproduces the reference bit for bit.  "cr": exp and
# log1p in double, rounded to float (correctly rounded results) -- what the HIP kernel computes; it
# differs from glibc's log1pf by one ulp in a small fraction of arguments.
MATH = "libm"
_LIBM = None


def _libm():
    global _LIBM
    if _LIBM is None:
        _LIBM = ctypes.CDLL("libm.so.6")
        for fn in (_LIBM.expf, _LIBM.log1pf):
            fn.restype = ctypes.c_float
            fn.argtypes = [ctypes.c_float]
    return _LIBM


def logsumexpf(x, y):
    """c_hashdecode.c:50-54: max + (|d| < 17 ? log1pf(expf(-|d|)) : 0), float32 throughout."""
    x, y = f32(x), f32(y)
    absdif = f32(abs(f32(x - y)))
    if not absdif < f32(17.0):
        return f32(max(x, y))
    if MATH == "libm":
        m = _libm()
        tail = f32(m.log1pf(m.expf(float(-absdif))))
    else:
        tail = f32(np.log1p(np.float64(f32(np.exp(-np.float64(absdif))))))
    return f32(max(x, y) + tail)


def backward(score, init=None):
    """c_flipflopfwdbwd.c:55-91 / decodeutil.pyx:55-81: (bwd (T+1, 2nb), total)."""
    score = np.asarray(score, dtype=f32)
    T, ntrans = score.shape
    nb = int(round((np.sqrt(1 + 2 * ntrans) - 1) / 2))
    ns = 2 * nb
    bwd = np.zeros((T + 1, ns), dtype=f32)
    if init is not None:
        bwd[T] = init
    for blk in range(T, 0, -1):
        p, c, s = bwd[blk], bwd[blk - 1], score[blk - 1]
        for b in range(nb):
            c[b] = f32(s[ns * nb + b] + p[nb + b])
            c[b + nb] = f32(s[ns * nb + b + nb] + p[nb + b])
        for to in range(nb):
            for fr in range(ns):
                c[fr] = logsumexpf(c[fr], f32(s[to * ns + fr] + p[to]))
    total = bwd[0, 0]
    for i in range(1, ns):
        total = logsumexpf(total, bwd[0, i])
    return bwd, float(total)


def qsort_inplace(A, less):
    """qsort.h:39-186 on the list A with LESS(i, j) = less(A[i], A[j]): the same comparisons and swaps
    in the same order, so elements that compare equal end where the reference leaves them."""
    n = len(A)
    if n <= 1:
        return A

    def LESS(i, j):
        return less(A[i], A[j])

    def SWAP(i, j):
        A[i], A[j] = A[j], A[i]

    def sort3(a1, a2, a3):                                  # qsort.h:41-57
        if LESS(a2, a1):
            if LESS(a3, a2):
                SWAP(a1, a3)
            else:
                SWAP(a1, a2)
                if LESS(a3, a2):
                    SWAP(a2, a3)
        elif LESS(a3, a2):
            SWAP(a2, a3)
            if LESS(a2, a1):
                SWAP(a1, a2)

    lo, hi, stack = 0, n - 1, []
    while True:
        if hi - lo + 1 >= 16:                               # Q_THRESH, qsort.h:112
            m = lo + ((hi - lo) >> 1)                       # partition, qsort.h:62-92
            sort3(lo + 1, m, hi)
            SWAP(lo, m)
            i, j = lo + 1, hi
            while True:
                i += 1
                while LESS(i, lo):
                    i += 1
                j -= 1
                while LESS(lo, j):
                    j -= 1
                if i >= j:
                    break
                SWAP(i, j)
            i = j + 1
            SWAP(lo, j)
            j -= 1
            # subfiles [lo, j] and [i, hi]: the larger is pushed, the smaller goes next; a subfile of
            # one element needs nothing (qsort.h:150-172)
            if j - lo >= hi - i:
                big, small = (lo, j), (i, hi)
            else:
                big, small = (i, hi), (lo, j)
            if small[0] == small[1]:
                lo, hi = big
            else:
                stack.append(big)
                lo, hi = small
        else:
            for q in range(lo + 1, hi + 1):                 # insertion sort, qsort.h:101-108
                k = q
                while k > lo and LESS(k, k - 1):
                    SWAP(k, k - 1)
                    k -= 1
            if not stack:
                break
            lo, hi = stack.pop()
    return A


def forward(score, init=None):
    """c_flipflopfwdbwd.c:112-152 / decodeutil.pyx:82-108: (fwd (T+1, 2nb), total)."""
    score = np.asarray(score, dtype=f32)
    T, ntrans = score.shape
    nb = int(round((np.sqrt(1 + 2 * ntrans) - 1) / 2))
    ns = 2 * nb
    fwd = np.zeros((T + 1, ns), dtype=f32)
    if init is not None:
        fwd[0] = init
    for blk in range(T):
        p, c, s = fwd[blk], fwd[blk + 1], score[blk]
        for b in range(nb):
            c[b + nb] = logsumexpf(f32(s[ns * nb + b] + p[b]), f32(s[ns * nb + b + nb] + p[b + nb]))
        for to in range(nb):
            c[to] = f32(s[to * ns] + p[0])
            for fr in range(1, ns):
                c[to] = logsumexpf(c[to], f32(s[to * ns + fr] + p[fr]))
    total = fwd[T, 0]
    for i in range(1, ns):
        total = logsumexpf(total, fwd[T, i])
    return fwd, float(total)


def beamsearch(score, beam_cut=0.0, beam_width=5, guided=True):
    """decodeutil.pyx:9-51 + c_hashdecode.c:346-507.  Returns (sequence int8 (flip-flop states), score)."""
    score = np.ascontiguousarray(score, dtype=f32)
    T, ntrans = score.shape
    nb = int(round((np.sqrt(1 + 2 * ntrans) - 1) / 2))
    ns = 2 * nb
    bwd = backward(score)[0] if guided else np.zeros((T + 1, ns), dtype=f32)
    with np.errstate(divide="ignore"):
        logcut = f32(np.log(f32(beam_cut)))
    seed = 0x880355F21E6D1965

    def move_idx(fr, to):
        return fr + 2 * nb * (to if to < nb else nb)

    beam = [dict(seq=[i], hash=chainfasthash64(seed, i), score=f32(0.0)) for i in range(nb)]
    for blk in range(T):
        cs, bs = score[blk], bwd[blk + 1]
        prev = beam
        pb = prev[0]["seq"][-1]
        mx = f32(cs[nb * ns + pb] + bs[pb + nb if pb < nb else pb])
        for i in range(nb):
            mx = f32(max(mx, f32(cs[i * ns + pb] + bs[i])))
        mx = f32(mx + prev[0]["score"])
        recs = []
        for i, el in enumerate(prev):
            pbase = el["seq"][-1]
            for base in range(nb):
                nbse = base if base != pbase else pbase + nb
                sc = f32(f32(el["score"] + cs[move_idx(pbase, nbse)]) + bs[nbse])
                if sc < f32(mx + logcut):
                    continue
                mx = max(mx, sc)
                recs.append(dict(hash=chainfasthash64(el["hash"], nbse), base=nbse, score=sc, orig=i))
        for i, el in enumerate(prev):
            base = el["seq"][-1]
            sc = f32(f32(el["score"] + cs[move_idx(base, base)]) + bs[base])
            if sc < f32(mx + logcut):
                continue
            mx = max(mx, sc)
            recs.append(dict(hash=el["hash"], base=-1, score=sc, orig=i))
        # merge records of the same sequence: sort by hash, fold runs of equal hash into their first
        # record (c_hashdecode.c:453-470), then sort by score (:472)
        qsort_inplace(recs, lambda x, y: x["hash"] > y["hash"])
        nuniq, j = (1 if recs else 0), 0
        for i in range(1, len(recs)):
            if recs[i]["hash"] == recs[j]["hash"]:
                recs[j]["score"] = logsumexpf(recs[i]["score"], recs[j]["score"])
                recs[i]["score"] = f32(-np.inf)
            else:
                j = i
                nuniq += 1
        qsort_inplace(recs, lambda x, y: x["score"] > y["score"])
        beam = []
        for r in recs[:min(beam_width, nuniq)]:        # c_hashdecode.c:474
            el = prev[r["orig"]]
            seq = list(el["seq"])
            h = el["hash"]
            if r["base"] != -1:
                seq.append(r["base"])
                h = r["hash"]
            beam.append(dict(seq=seq, hash=h, score=f32(r["score"] - bs[seq[-1]])))
    best = beam[0]
    return np.array(best["seq"][:T], dtype=np.int8), float(best["score"])


# ---- the genuine reference (oracle/_ref/libref_decodeutil.so) ------------------------------
_REF = None


def ref_available():
    return os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_decodeutil.so"))


def ref_beamsearch(score, beam_cut=0.0, beam_width=5, guided=True):
    """decodeutil.beamsearch through the reference C, wrapper logic of decodeutil.pyx:36-51."""
    global _REF
    if _REF is None:
        _REF = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_decodeutil.so"))
        _REF.flipflop_beamsearch.restype = ctypes.c_float
        _REF.flipflop_backward.restype = ctypes.c_float
    score = np.ascontiguousarray(score, dtype=f32)
    T, ntrans = score.shape
    nb = int(round((np.sqrt(1 + 2 * ntrans) - 1) / 2))
    bwd = np.zeros((T + 1, 2 * nb), dtype=f32)
    fp = ctypes.POINTER(ctypes.c_float)
    if guided:
        _REF.flipflop_backward(score.ctypes.data_as(fp), ctypes.c_size_t(nb), ctypes.c_size_t(T), bwd.ctypes.data_as(fp))
    res = np.zeros(T + 8, dtype=np.int8)
    sc = _REF.flipflop_beamsearch(score.ctypes.data_as(fp), ctypes.c_size_t(nb), ctypes.c_size_t(T),
                                  bwd.ctypes.data_as(fp), ctypes.c_int(int(beam_width)), ctypes.c_float(beam_cut),
                                  res.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)))
    res = res[:T]
    neg = np.nonzero(res == -1)[0]
    return res[:neg[0]] if len(neg) else res, float(sc), bwd


def ref_lattice(score, init=None, forward_pass=True):
    """decodeutil.forward / backward through the reference C (wrapper logic of decodeutil.pyx:54-108)."""
    ref_beamsearch(np.zeros((1, 40), dtype=f32))          # loads the library
    score = np.ascontiguousarray(score, dtype=f32)
    T, ntrans = score.shape
    nb = int(round((np.sqrt(1 + 2 * ntrans) - 1) / 2))
    res = np.zeros((T + 1, 2 * nb), dtype=f32)
    if init is not None:
        res[0 if forward_pass else T] = init
    fp = ctypes.POINTER(ctypes.c_float)
    _REF.flipflop_forward.restype = ctypes.c_float
    fn = _REF.flipflop_forward if forward_pass else _REF.flipflop_backward
    tot = fn(score.ctypes.data_as(fp), ctypes.c_size_t(nb), ctypes.c_size_t(T), res.ctypes.data_as(fp))
    return res, float(tot)
