"""oracle -- CPU checker for the flip-flop CRF hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; nothing under ``taiyaki_amd/`` does.  It wraps

* ``liboracle.so``  -- the repo's own plain-C restatement (``flipflop_oracle.c``);
* ``_ref/libref_ctc.so`` -- the genuine reference C, compiled from the sources
  where they lie under ``/root/reference`` by ``oracle/Makefile`` (optional; present
  in the build container and shipped prebuilt to the GPU box).

The numpy-level functions restate the reference's Python wrappers
(``taiyaki/ctc/ctc.pyx``) so a test reads like the reference's own test.
Parity status: PINNED (see flipflop_oracle.c header and tests/test_oracle_pinning.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_szp = ctypes.POINTER(ctypes.c_size_t)
_sz = ctypes.c_size_t


def build(quiet=True):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_ctc.so"))


def ref():
    """The genuine reference C library (crf_flipflop_cost/grad, cat_mod_...)."""
    global _REF
    if _REF is None:
        _REF = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_ctc.so"))
    return _REF


def set_threads(n):
    """Thread count for the OpenMP loops over reads (both libraries)."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(n))
    except OSError:
        pass


# ---------------------------------------------------------------------------
# index algebra (flipflopfings.py:6-78)
# ---------------------------------------------------------------------------
def nbase_flipflop(nstate):
    """flipflopfings.py:171-184"""
    nbase_f = np.sqrt(0.25 + (0.5 * np.float32(nstate))) - 0.5
    assert np.mod(nbase_f, 1) == 0, "Number of states not valid for flip-flop model"
    return int(np.round(nbase_f))


def flipflop_code(bases, nbase=4):
    bases = np.ascontiguousarray(bases, dtype=np.int32)
    out = np.empty_like(bases)
    lib().oracle_flipflop_code(_ptr(bases, _i32p), _sz(len(bases)), _sz(nbase),
                               _ptr(out, _i32p))
    return out


def flipflop_indices(seqs, seqlen, nbase):
    """(moveidxs, stayidxs) as the concatenated uintp arrays ctc.pyx:127-134 builds."""
    seqs = np.ascontiguousarray(seqs, dtype=np.int32)
    seqlen = np.ascontiguousarray(seqlen, dtype=np.int32)
    nbatch = len(seqlen)
    nmove = max(int(seqlen.sum()) - int((seqlen > 0).sum()), 0)
    # the reference has exactly (sum(seqlen) - nbatch) moves; a zero-length read
    # would make that inconsistent, so (like np.split there) we keep its layout
    move = np.zeros(max(int(seqlen.sum()) - nbatch, 0) + nbatch, dtype=np.uintp)
    stay = np.zeros(max(int(seqlen.sum()), 1), dtype=np.uintp)
    lib().oracle_flipflop_indices(_ptr(seqs, _i32p), _ptr(seqlen, _i32p),
                                  _sz(nbatch), _sz(nbase), _ptr(move, _szp),
                                  _ptr(stay, _szp))
    del nmove
    return move, stay


# ---------------------------------------------------------------------------
# (A) sequence-constrained CRF, numpy-level (ctc.pyx:31-113, 162-255)
# ---------------------------------------------------------------------------
def _seq_call(libobj, prefix, logprob, move, stay, seqlen, modmove=None,
              modfact=None, want_grad=True):
    logprob = np.ascontiguousarray(logprob, dtype=np.float32)
    assert np.all(np.isfinite(logprob)), "Input not finite"     # ctc.pyx:48
    nblk, nbatch, nstate = logprob.shape
    seqlen = np.ascontiguousarray(seqlen, dtype=np.int32)
    move = np.ascontiguousarray(move, dtype=np.uintp)
    stay = np.ascontiguousarray(stay, dtype=np.uintp)
    costs = np.zeros(nbatch, dtype=np.float32)
    grads = np.zeros_like(logprob) if want_grad else None
    args = [_ptr(logprob, _f32p), _sz(nstate), _sz(nblk), _sz(nbatch),
            _ptr(move, _szp), _ptr(stay, _szp)]
    if modmove is not None:
        modmove = np.ascontiguousarray(modmove, dtype=np.uintp)
        modfact = np.ascontiguousarray(modfact, dtype=np.float32)
        args += [_ptr(modmove, _szp), _ptr(modfact, _f32p)]
        name = prefix + "cat_mod_flipflop_"
    else:
        name = prefix + "crf_flipflop_"
    args += [_ptr(seqlen, _i32p), _ptr(costs, _f32p)]
    if want_grad:
        args.append(_ptr(grads, _f32p))
        fn = getattr(libobj, name + "grad")
    else:
        fn = getattr(libobj, name + "cost")
    fn.restype = None
    fn(*args)
    assert np.all(np.isfinite(costs)), "Error: all costs must be finite"
    if want_grad:
        assert np.all(np.isfinite(grads)), "Error: Gradients not finite"
        return -costs / nblk, -grads / nblk     # ctc.pyx:113
    return -costs / nblk, None                  # ctc.pyx:66


def _empty_reads_apart(fn):
    """An EMPTY read that is not the batch's last shifts the move indices of every read after it by one in the
    reference: ctc.pyx:127-129 emits no move for it while c_crf_flipflop.c:479 (`moveidxs + seqidx[batch] -
    batch`) counts seqlen - 1 = -1 -- later reads are scored with a neighbour's transition, which nobody means
    (its data pipeline never emits such a batch).  The restated C keeps that pointer arithmetic; this wrapper
    keeps such batches well-defined: the live reads are evaluated as a batch of their own (each with its own
    indices, the reference's result for every read of a batch without empty reads), the empty ones get the
    reference's cost 0 / zero gradient rows (c_crf_flipflop.c:458-464).  DESIGN.md lists it as a deviation."""
    import functools

    @functools.wraps(fn)
    def wrapped(logprob, seqs, seqlen, *args, **kw):
        sl = np.asarray(seqlen)
        live = sl > 0
        if live.all() or not live.any() or not live[np.argmin(live):].any():
            return fn(logprob, seqs, seqlen, *args, **kw)       # (no empty read, or only trailing ones)
        logprob = np.asarray(logprob)
        cost, grads = fn(np.ascontiguousarray(logprob[:, live, :]), seqs, np.ascontiguousarray(sl[live]), *args, **kw)
        full_cost = np.zeros(len(sl), dtype=cost.dtype)
        full_cost[live] = cost
        full_grads = None
        if grads is not None:
            full_grads = np.zeros(logprob.shape, dtype=grads.dtype)
            full_grads[:, live, :] = grads
        return full_cost, full_grads
    return wrapped


@_empty_reads_apart
def crf_flipflop_loss(logprob, seqs, seqlen, sharpfact=1.0, want_grad=True,
                      use_ref=False):
    """FlipFlopCRF.forward semantics (ctc.pyx:116-151) on numpy arrays.

    Returns (loss (N,), dloss/dlogprob (T,N,S) or None).  The saved gradient of
    the reference is -grad_C/T w.r.t. lp = sharp*logprob; sharp and 1/sharp
    cancel so it is exactly d out / d logprob.
    """
    logprob = np.asarray(logprob, dtype=np.float32)
    lp = (np.float32(sharpfact) * logprob).astype(np.float32)
    nbase = nbase_flipflop(lp.shape[2])
    move, stay = flipflop_indices(seqs, seqlen, nbase)
    libobj, prefix = (ref(), "") if use_ref else (lib(), "oracle_")
    cost, grads = _seq_call(libobj, prefix, lp, move, stay, seqlen,
                            want_grad=want_grad)
    return (cost / np.float32(sharpfact)).astype(np.float32), grads


@_empty_reads_apart
def crf_flipflop_loss_f64(logprob, seqs, seqlen, sharpfact=1.0, mod_cats=None, can_mods_offsets=None,
                          mod_cat_weights=None):
    """The FLOAT64 WITNESS (flipflop_oracle.c: oracle_seq_grad_f64) behind the same operator
    semantics as `crf_flipflop_loss` / `cat_mod_flipflop_loss`: (loss (N,), dloss/dlogprob
    (T, N, S)) as float64 arrays.  What the fp32 reference and the HIP kernels both approximate;
    used where the reference's own rounding noise (long T, wild cat-mod logits) exceeds the
    tolerance the kernels are held to."""
    logprob = np.asarray(logprob, dtype=np.float32)
    ntrans = logprob.shape[2]
    nmod = int(np.asarray(can_mods_offsets)[-1]) if mod_cats is not None else 0
    nbase = nbase_flipflop(ntrans - nmod)
    sharp = np.ones(ntrans, dtype=np.float32)
    sharp[:ntrans - nmod] = np.float32(sharpfact)               # ctc.pyx:119 / 265-267
    lp = np.ascontiguousarray(logprob * sharp, dtype=np.float32)
    move, stay = flipflop_indices(seqs, seqlen, nbase)
    seqlen = np.ascontiguousarray(seqlen, dtype=np.int32)
    nblk, nbatch, _ = lp.shape
    mm = mf = None
    if mod_cats is not None:
        mm, mf = cat_mod_indices(seqs, seqlen, mod_cats, can_mods_offsets, mod_cat_weights, nbase)
        mm = np.ascontiguousarray(mm, dtype=np.uintp)
        mf = np.ascontiguousarray(mf, dtype=np.float32)
    score = np.zeros(nbatch, dtype=np.float64)
    grad = np.zeros(lp.shape, dtype=np.float64)
    _f64p = ctypes.POINTER(ctypes.c_double)
    fn = lib().oracle_seq_grad_f64
    fn.restype = None
    fn(_ptr(lp, _f32p), _sz(ntrans), _sz(nblk), _sz(nbatch), _ptr(move, _szp), _ptr(stay, _szp),
       _ptr(mm, _szp) if mm is not None else None, _ptr(mf, _f32p) if mf is not None else None,
       _ptr(seqlen, _i32p), _ptr(score, _f64p), _ptr(grad, _f64p))
    # cost = -score / nblk / sharp; the saved gradient is d cost / d lp (cat-mod: unscaled, ctc.pyx:306-310;
    # plain: sharp and 1 / sharp cancel)
    return -score / nblk / float(sharpfact), -grad / nblk


def cat_mod_indices(seqs, seqlen, mod_cats, can_mods_offsets, mod_cat_weights,
                    nbase):
    """ctc.pyx:282-292: modmoveidxs, modmovefacts (concatenated, one per move)."""
    seqs = np.asarray(seqs, dtype=np.int64)
    mod_cats = np.asarray(mod_cats, dtype=np.int64)
    seqlen = np.asarray(seqlen, dtype=np.int64)
    can_mods_offsets = np.asarray(can_mods_offsets, dtype=np.int64)
    mod_cat_weights = np.asarray(mod_cat_weights, dtype=np.float32)
    mod_offset = (nbase + 1) * nbase * 2
    starts = np.concatenate([[0], np.cumsum(seqlen)])
    pieces = []
    for b in range(len(seqlen)):
        s = seqs[starts[b]:starts[b + 1]]
        m = mod_cats[starts[b]:starts[b + 1]]
        pieces.append(can_mods_offsets[np.mod(s[1:], nbase)] + m[1:])
    mod_seq = (np.concatenate(pieces) if pieces else np.zeros(0)).astype(np.int64)
    modmoveidxs = (mod_offset + mod_seq).astype(np.uintp)
    modmovefacts = mod_cat_weights[mod_seq].astype(np.float32)
    if len(modmoveidxs) == 0:
        modmoveidxs = np.zeros(1, dtype=np.uintp)
        modmovefacts = np.zeros(1, dtype=np.float32)
    return modmoveidxs, modmovefacts


@_empty_reads_apart
def cat_mod_flipflop_loss(logprob, seqs, seqlen, mod_cats, can_mods_offsets,
                          mod_cat_weights, sharpfact=1.0, want_grad=True,
                          use_ref=False):
    """CatModFlipFlop.forward semantics (ctc.pyx:258-310).

    Quirk reproduced: only the canonical columns are sharpened (265-267) and the
    saved gradient is returned unscaled (306-310), i.e. it is d cost / d lp.
    """
    logprob = np.asarray(logprob, dtype=np.float32)
    ntrans = logprob.shape[2]
    n_can_trans = ntrans - int(np.asarray(can_mods_offsets)[-1])
    nbase = nbase_flipflop(n_can_trans)
    trans_sharp = np.ones(ntrans, dtype=np.float32)
    trans_sharp[:n_can_trans] = sharpfact
    lp = np.ascontiguousarray(logprob * trans_sharp, dtype=np.float32)
    move, stay = flipflop_indices(seqs, seqlen, nbase)
    modmove, modfact = cat_mod_indices(seqs, seqlen, mod_cats, can_mods_offsets,
                                       mod_cat_weights, nbase)
    libobj, prefix = (ref(), "") if use_ref else (lib(), "oracle_")
    cost, grads = _seq_call(libobj, prefix, lp, move, stay, seqlen, modmove,
                            modfact, want_grad=want_grad)
    return (cost / np.float32(sharpfact)).astype(np.float32), grads


# ---------------------------------------------------------------------------
# (B) log-partition, posterior; Viterbi
# ---------------------------------------------------------------------------
def flipflop_logz(scores):
    """layers.log_partition_flipflop(scores).squeeze(1) (layers.py:1277-1299)."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    T, N, S = scores.shape
    out = np.zeros(N, dtype=np.float32)
    lib().oracle_flipflop_logz(_ptr(scores, _f32p), _sz(T), _sz(N),
                               _sz(nbase_flipflop(S)), _ptr(out, _f32p))
    return out


def flipflop_logz_grad(scores):
    """(logZ (N,), d logZ / d scores (T,N,S)) -- cupy_extensions/flipflop.py:338-368;
    the gradient is also decode.flipflop_make_trans (decode.py:42-72)."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    T, N, S = scores.shape
    out = np.zeros(N, dtype=np.float32)
    grad = np.zeros_like(scores)
    lib().oracle_flipflop_logz_grad(_ptr(scores, _f32p), _sz(T), _sz(N),
                                    _sz(nbase_flipflop(S)), _ptr(out, _f32p),
                                    _ptr(grad, _f32p))
    return out, grad


def flipflop_viterbi(scores):
    """decode._flipflop_viterbi (decode.py:75-115): fwd, traceback, path."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    T, N, S = scores.shape
    nb = nbase_flipflop(S)
    fwd = np.zeros((T + 1, N, 2 * nb), dtype=np.float32)
    tb = np.zeros((T, N, 2 * nb), dtype=np.int64)
    path = np.zeros((T + 1, N), dtype=np.int64)
    lib().oracle_flipflop_viterbi(_ptr(scores, _f32p), _sz(T), _sz(N), _sz(nb),
                                  _ptr(fwd, _f32p), _ptr(tb, _i64p),
                                  _ptr(path, _i64p))
    return fwd, tb, path


# ---------------------------------------------------------------------------
# basecall-side consumers: error probabilities, chunk stitching
# ---------------------------------------------------------------------------
def errprobs_from_trans(trans, path):
    """qscores.errprobs_from_trans (qscores.py:88-142), numpy restatement.
    baseprobs[b] = sum of the posterior weights of every transition into base b -- the 2nb
    transitions into b_flip, b_flip -> b_flop, b_flop stay (qscores.py:58-85) -- normalised
    by (their sum over b + SMALL_VAL = 1e-10, constants.py:7); errprob = 1 - baseprobs at
    path[t + 1] % nb; row 0 = 1 - 2.0."""
    trans = np.asarray(trans, dtype=np.float32)
    path = np.asarray(path)
    T, N, S = trans.shape
    nb = nbase_flipflop(S)
    base = np.zeros((T, N, nb), dtype=np.float32)
    for b in range(nb):
        idx = list(range(2 * nb * b, 2 * nb * (b + 1))) + [2 * nb * nb + b, 2 * nb * nb + nb + b]
        base[:, :, b] = trans[:, :, idx].sum(axis=2, dtype=np.float32)
    base = base / (base.sum(axis=2, keepdims=True, dtype=np.float32) + np.float32(1e-10))
    p = np.empty((T + 1, N), dtype=np.float32)
    p[1:] = np.take_along_axis(base, (path[1:] % nb)[:, :, None], axis=2)[:, :, 0]
    p[0] = 2.0
    return (np.float32(1.0) - p).astype(np.float32)


def stitch_chunks(out, chunk_starts, chunk_ends, stride, path_stitching=False):
    """basecall_helpers.stitch_chunks (basecall_helpers.py:46-94), numpy restatement:
    first / middle / last chunk cut at the midpoints of the overlaps."""
    out = np.asarray(out)
    nchunks = out.shape[1]
    if nchunks == 1:
        return out[:, 0]
    one = 1 if path_stitching else 0
    start = chunk_starts[0] // stride
    end = (chunk_ends[0] + chunk_starts[1]) // (2 * stride) + one
    parts = [out[start:end, 0]]
    for i in range(1, nchunks - 1):
        start = (chunk_ends[i - 1] - chunk_starts[i]) // (2 * stride) + one
        end = (chunk_ends[i] + chunk_starts[i + 1] - 2 * chunk_starts[i]) // (2 * stride) + one
        parts.append(out[start:end, i])
    start = (chunk_ends[-2] - chunk_starts[-1]) // (2 * stride) + one
    end = (chunk_ends[-1] - chunk_starts[-1]) // stride + one
    parts.append(out[start:end, -1])
    return np.concatenate(parts, 0)
