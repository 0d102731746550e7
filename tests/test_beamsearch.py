"""Hash beam search (taiyaki/decodeutil/c_hashdecode.c:346-507).

CPU: the oracle restatement (oracle/beam.py) against the genuine reference C compiled into
oracle/_ref/libref_decodeutil.so and against the golden fixtures it produced
(tests/golden/make_golden_beam.py).  GPU (-m gpu): the HIP kernel against the fixtures, the
restatement and -- where oracle/_ref travelled -- the reference C itself."""
import numpy as np
import pytest

from oracle import beam
from tests.conftest import load_golden
from tests.golden import make_golden_beam as mk


def test_hash_chain_known_values():
    """fasthash.c:95-103 on hand-checked values (seed = the multiplier, as c_hashdecode.c:353)."""
    h = beam.chainfasthash64(0x880355F21E6D1965, 0)
    assert h == beam.chainfasthash64(0x880355F21E6D1965, 0) and h != beam.chainfasthash64(0x880355F21E6D1965, 1)
    assert 0 <= h < 1 << 64
    # appending the same states in the same order gives the same hash, another order another one
    a = beam.chainfasthash64(beam.chainfasthash64(h, 2), 5)
    assert a == beam.chainfasthash64(beam.chainfasthash64(h, 2), 5) != beam.chainfasthash64(beam.chainfasthash64(h, 5), 2)


@pytest.mark.parametrize("name", ["t40w5", "t300w5", "t300w5_sharp", "t200w12", "t250w3_unguided", "t250w5_cut",
                                  "t120w1"])
def test_restatement_matches_reference_goldens(name):
    gold = load_golden("beam_small.npz")
    spec = mk.CASES[name]
    seq, score = beam.beamsearch(mk.case_scores(spec), spec[4], spec[3], spec[5])
    assert np.array_equal(seq, gold[name + "/seq"])
    assert np.float32(score) == gold[name + "/score"]           # bit for bit (glibc's expf / log1pf)


def quantised_scores(T, seed, step=0.5):
    """Scores on a grid of `step`: different paths then reach EXACTLY equal scores all the time, and
    the beam depends on the order the reference's quicksort leaves equal records in."""
    rng = np.random.RandomState(seed)
    return (np.round(rng.randn(T, 40) * 2 / step) * step).astype(np.float32)


def test_qsort_restatement_orders_equal_keys_like_the_macro():
    """qsort.h on few distinct keys: below 16 records it is an insertion sort (stable); from 16 on
    the partition step moves equal keys around -- checked against the reference library through
    whole beam searches below, here for being a sort at all and for the two regimes."""
    rng = np.random.RandomState(0)
    for n in (1, 2, 5, 15, 16, 17, 25, 40, 60):
        for _ in range(20):
            keys = [(float(rng.randint(0, 4)), k) for k in range(n)]
            out = beam.qsort_inplace(list(keys), lambda x, y: x[0] > y[0])
            assert sorted(out) == sorted(keys) and all(out[i][0] >= out[i + 1][0] for i in range(n - 1))
            if n < 16:
                assert out == sorted(keys, key=lambda kv: -kv[0])          # stable
    keys = [(1.0, k) for k in range(20)]
    assert beam.qsort_inplace(list(keys), lambda x, y: x[0] > y[0]) != keys     # the partition is not stable


def test_restatement_follows_the_reference_through_exact_ties():
    """t800w5 has four blocks with tied scores at the edge of the beam; the quantised cases have
    them in nearly every block.  Sequences and scores bit for bit."""
    gold = load_golden("beam_small.npz")
    spec = mk.CASES["t800w5"]
    seq, score = beam.beamsearch(mk.case_scores(spec), spec[4], spec[3], spec[5])
    assert np.array_equal(seq, gold["t800w5/seq"]) and np.float32(score) == gold["t800w5/score"]
    for name in ("q150w5", "q120w12", "q90w7_cut"):
        T, seed, w, cut, g = mk.QUANTISED[name]
        seq, score = beam.beamsearch(quantised_scores(T, seed), cut, w, g)
        assert np.array_equal(seq, gold[name + "/seq"]) and np.float32(score) == gold[name + "/score"], name


def test_restatement_matches_the_reference_library_on_random_cases():
    if not beam.ref_available():
        pytest.skip("oracle/_ref/libref_decodeutil.so not built (no /root/reference here)")
    from taiyaki_amd import synth
    rng = np.random.RandomState(4)
    for k in range(10):
        T = int(rng.randint(5, 160))
        sc = (synth.scores(T, 1, 40, 900 + k)[:, 0, :] * np.float32(rng.uniform(0.3, 1.6))).astype(np.float32)
        w, cut, g = int(rng.randint(1, 13)), float(rng.choice([0.0, 0.0, 0.01, 0.2])), bool(rng.randint(2))
        rs, rsc, rbwd = beam.ref_beamsearch(sc, cut, w, g)
        ms, msc = beam.beamsearch(sc, cut, w, g)
        assert np.array_equal(rs, ms), (T, w, cut, g)
        assert np.float32(rsc) == np.float32(msc)
        if g:
            assert np.array_equal(beam.backward(sc)[0], rbwd)
    for k in range(6):                                      # tie-heavy
        w, cut, g = int(rng.choice([3, 5, 12])), float(rng.choice([0.0, 0.05])), bool(rng.randint(2))
        sc = quantised_scores(int(rng.randint(20, 140)), 50 + k)
        rs, rsc, _ = beam.ref_beamsearch(sc, cut, w, g)
        ms, msc = beam.beamsearch(sc, cut, w, g)
        assert np.array_equal(rs, ms) and np.float32(rsc) == np.float32(msc), (k, w, cut, g)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mk.CASES))
def test_hip_beamsearch_against_reference_goldens(gpu_device, name):
    from taiyaki_amd import decodeutil
    gold = load_golden("beam_small.npz")
    spec = mk.CASES[name]
    seq, score = decodeutil.beamsearch(mk.case_scores(spec), spec[4], spec[3], spec[5])
    assert seq.dtype == np.int8 and np.array_equal(seq, gold[name + "/seq"])
    # the kernel rounds exp / log1p correctly; glibc's log1pf is off by an ulp now and then
    assert abs(score - float(gold[name + "/score"])) <= 2e-6 * abs(score)


@pytest.fixture
def correctly_rounded_oracle():
    old, beam.MATH = beam.MATH, "cr"
    yield beam
    beam.MATH = old


@pytest.mark.gpu
def test_hip_beamsearch_orders_exact_ties_like_the_reference(gpu_device, correctly_rounded_oracle):
    """Scores on a grid: ties in nearly every block, so every block goes through the kernel's
    restatement of the reference's sort procedure.  Against the oracle with the kernel's rounding of
    exp / log1p (an ulp decides a tie here), which the CPU tests pin to the reference library."""
    import torch
    from taiyaki_amd import decodeutil
    cases = [(150, 31, 5, 0.0, True), (120, 32, 12, 0.0, True), (90, 33, 7, 0.05, True), (200, 34, 5, 0.0, False),
             (64, 35, 3, 0.0, True), (170, 36, 4, 0.0, True)]
    for T, seed, w, cut, g in cases:
        sc = quantised_scores(T, seed)
        ws, wsc = correctly_rounded_oracle.beamsearch(sc, cut, w, g)
        seq, score = decodeutil.beamsearch(torch.from_numpy(sc).to(gpu_device), cut, w, g)
        assert np.array_equal(seq, ws), (T, seed, w, cut, g)
        assert np.float32(score) == np.float32(wsc), (T, seed, w, cut, g)


@pytest.mark.gpu
def test_hip_beamsearch_batch_against_restatement(gpu_device):
    """A batch of reads in one launch (one wavefront per read) = each read on its own; against
    the restatement and, where present, the reference library."""
    import torch
    from taiyaki_amd import decodeutil, synth
    T, N = 180, 9
    sc = (synth.scores(T, N, 40, 77) * np.float32(0.8)).astype(np.float32)
    seqs, scores = decodeutil.beamsearch(torch.from_numpy(sc).to(gpu_device), 0.0, 5, True)
    assert len(seqs) == N and scores.shape == (N,)
    for n in range(N):
        one = np.ascontiguousarray(sc[:, n])
        ws, wsc = beam.beamsearch(one, 0.0, 5, True)
        assert np.array_equal(seqs[n], ws) and abs(scores[n] - wsc) <= 2e-6 * abs(wsc)
        if beam.ref_available():
            rs, rsc, _ = beam.ref_beamsearch(one, 0.0, 5, True)
            assert np.array_equal(seqs[n], rs)
    # decoded states are a valid flip-flop path: consecutive states differ, a repeated base flips
    for s in seqs:
        assert np.all(s[1:] != s[:-1]) and s.min() >= 0 and s.max() < 8


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(3))
def test_fuzz_beam_seeded_subset(gpu_device, correctly_rounded_oracle, block):
    """A bounded, seeded part of tests/helpers/fuzz_beam.py (150 cases per run when called as a
    script): random T / width / cut / guided over continuous, grid and saturated scores, batches of
    reads per launch; sequences and float scores bit for bit."""
    from tests.helpers import fuzz_beam
    rng = np.random.RandomState(500 + block)
    for k in range(8):
        ok, msg = fuzz_beam.case(k, rng, gpu_device)
        assert ok, msg


@pytest.mark.gpu
def test_hip_beamsearch_long_read_walks_back_through_global_memory(gpu_device):
    """Reads longer than the 3584 back-pointer rows that fit the LDS keep the table in HBM only
    (csrc/beam_kernels.hip: lds_rows); two reads in the launch, the short one padded with -1."""
    import torch
    from taiyaki_amd import decodeutil, synth
    T = 3700
    sc = (synth.scores(T, 2, 40, 91) * np.float32(0.8)).astype(np.float32)
    seqs, scores = decodeutil.beamsearch(torch.from_numpy(sc).to(gpu_device), 0.0, 5, True)
    for n in range(2):
        one = np.ascontiguousarray(sc[:, n])
        if beam.ref_available():
            ws, wsc, _ = beam.ref_beamsearch(one, 0.0, 5, True)
        else:
            ws, wsc = beam.beamsearch(one, 0.0, 5, True)
        assert np.array_equal(seqs[n], ws) and abs(scores[n] - wsc) <= 2e-6 * abs(wsc)


# ---- the decoder's lattice passes (decodeutil.forward / backward) -------------------------------
def reference_unit_test_weights():
    """test/unit/test_decodeutil.py:16-18: the reference's own fixture and known answer."""
    rs = np.random.RandomState(0xdeadbeef)
    return rs.randn(12, 40).astype("f4"), 27.16876983642578


def lse(x, axis=None):
    m = np.max(x, axis=axis, keepdims=True)
    return np.squeeze(m + np.log(np.sum(np.exp(x - m), axis=axis, keepdims=True)), axis=axis)


def check_reference_unit_test_properties(fwd_fn, bwd_fn):
    """test/unit/test_decodeutil.py:22-62 on the given implementation."""
    w, expt = reference_unit_test_weights()
    fwd, _ = fwd_fn(w)
    bwd, _ = bwd_fn(w)
    assert fwd.shape == bwd.shape == (13, 8)
    assert abs(float(lse(bwd[0])) - expt) < 1e-5 and abs(float(lse(fwd[-1])) - expt) < 1e-5      # :40-50
    score = lse(fwd + bwd, axis=1)                                                               # :52-62
    assert abs(float(score.mean()) - expt) < 1e-5 and float(score.max() - score.min()) < 1e-5
    init = np.zeros(8, dtype="f4")
    init[4:] = -50000                                                                            # :24-32
    fwd2, _ = fwd_fn(w, init=init)
    import oracle
    lz, _ = oracle.flipflop_logz_grad(w[:, None, :])
    assert abs(float(lse(fwd2[-1])) - float(lz[0])) < 1e-5
    assert abs(float(lse(bwd[0, :4])) - float(lz[0])) < 1e-5                                     # :34-38


def test_lattice_restatement_against_reference_library_and_unit_test():
    check_reference_unit_test_properties(beam.forward, beam.backward)
    if not beam.ref_available():
        pytest.skip("oracle/_ref/libref_decodeutil.so not built (no /root/reference here)")
    rng = np.random.RandomState(8)
    for k in range(6):
        sc = (rng.randn(int(rng.randint(1, 90)), 40) * 2).astype(np.float32)
        init = None if k % 2 else (rng.randn(8) * 3).astype(np.float32)
        for fwdp, fn in ((True, beam.forward), (False, beam.backward)):
            mine, ref = fn(sc, init), beam.ref_lattice(sc, init, fwdp)
            assert np.array_equal(mine[0], ref[0]) and np.float32(mine[1]) == np.float32(ref[1])


@pytest.mark.gpu
def test_hip_lattice_passes(gpu_device, correctly_rounded_oracle):
    """decodeutil.forward / backward on the device: the reference's unit test, the oracle with the
    kernel's rounding bit for bit (one read and a batch, with and without an initial vector), the
    reference library to float rounding."""
    import torch
    from taiyaki_amd import decodeutil
    check_reference_unit_test_properties(decodeutil.forward, decodeutil.backward)
    rng = np.random.RandomState(9)
    T, N = 150, 5
    sc = (rng.randn(T, N, 40) * 2).astype(np.float32)
    init = (rng.randn(N, 8) * 3).astype(np.float32)
    for fwdp, fn, ofn in ((True, decodeutil.forward, correctly_rounded_oracle.forward),
                          (False, decodeutil.backward, correctly_rounded_oracle.backward)):
        for use_init in (False, True):
            mats, tots = fn(torch.from_numpy(sc).to(gpu_device), init if use_init else None)
            assert mats.shape == (N, T + 1, 8) and tots.shape == (N,)
            for n in range(N):
                want = ofn(np.ascontiguousarray(sc[:, n]), init[n] if use_init else None)
                assert np.array_equal(mats[n], want[0]) and np.float32(tots[n]) == np.float32(want[1])
                if beam.ref_available():
                    ref = beam.ref_lattice(np.ascontiguousarray(sc[:, n]), init[n] if use_init else None, fwdp)
                    assert np.abs(mats[n] - ref[0]).max() <= 1e-5 * max(1.0, np.abs(ref[0]).max())
    one, tot = decodeutil.backward(sc[:, 0])                    # numpy in, one read: the reference's call
    assert one.shape == (T + 1, 8) and isinstance(tot, float)


@pytest.mark.gpu
@pytest.mark.parametrize("nbase", [1, 2, 3])
def test_hip_beamsearch_and_lattice_other_alphabets(gpu_device, correctly_rounded_oracle, nbase):
    """Alphabets of 1-3 bases (ntrans = 2 nbase (nbase + 1)): the kernels index by nbase, nothing is
    fixed to 40 columns."""
    import torch
    from taiyaki_amd import decodeutil
    rng = np.random.RandomState(40 + nbase)
    ntrans = 2 * nbase * (nbase + 1)
    for T, w, cut, guided in ((60, 5, 0.0, True), (45, 3, 0.05, True), (30, 12 if nbase < 4 else 8, 0.0, False)):
        sc = (rng.randn(T, ntrans) * 2).astype(np.float32)
        ws, wsc = correctly_rounded_oracle.beamsearch(sc, cut, w, guided)
        seq, score = decodeutil.beamsearch(torch.from_numpy(sc).to(gpu_device), cut, w, guided)
        assert np.array_equal(seq, ws) and np.float32(score) == np.float32(wsc), (nbase, T, w, cut, guided)
        if beam.ref_available():
            rs, rsc, _ = beam.ref_beamsearch(sc, cut, w, guided)
            assert np.array_equal(seq, rs) and abs(score - rsc) <= 2e-6 * max(1.0, abs(rsc))
        for fn, ofn in ((decodeutil.forward, correctly_rounded_oracle.forward),
                        (decodeutil.backward, correctly_rounded_oracle.backward)):
            mat, tot = fn(sc)
            want = ofn(sc)
            assert np.array_equal(mat, want[0]) and np.float32(tot) == np.float32(want[1])
