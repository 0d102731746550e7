"""Pin the CPU oracle (oracle/flipflop_oracle.c) before anything trusts it.

Checked against: the genuine reference C built into oracle/_ref (when present),
the reference's embedded known-answer harnesses, the reference's own unit-test
vectors, and golden fixtures produced by importing the reference Python
(tests/golden/make_golden.py).  All CPU; no GPU needed.
"""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import load_golden
from tests.golden import cases


def _check_grad(gold, prefix, grad, atol):
    if prefix in gold.files:
        np.testing.assert_allclose(grad, gold[prefix], atol=atol, rtol=0)
    else:
        cs = cases.grad_checksums(grad)
        np.testing.assert_allclose(cs["sum"], gold[prefix + "_sum"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(cs["sumsq"], gold[prefix + "_sumsq"], rtol=1e-4,
                                   atol=1e-7)
        np.testing.assert_array_equal(cs["sample_idx"], gold[prefix + "_sample_idx"])
        np.testing.assert_allclose(cs["sample"], gold[prefix + "_sample"], atol=atol,
                                   rtol=0)


# --------------------------------------------------------------------------
# known-answer tests held by the reference
# --------------------------------------------------------------------------
def test_c_harness_crf_twostate(oracle_mod):
    """c_crf_flipflop.c:520-766: forward == backward == -2.378088 for both reads."""
    ka = load_golden("known_answers.npz")
    lp = ka["ccrf/logprob"]
    move = ka["ccrf/move"].astype(np.uintp)
    stay = ka["ccrf/stay"].astype(np.uintp)
    seqlen = ka["ccrf/seqlen"]
    T = lp.shape[0]
    cost, grad = oracle_mod._seq_call(oracle_mod.lib(), "oracle_", lp, move, stay, seqlen)
    np.testing.assert_allclose(-cost * T, ka["ccrf/score"], atol=2e-6)
    # both batch elements are identical -> identical gradients (harness 'Max grad delta')
    np.testing.assert_allclose(grad[:, 0], grad[:, 1], atol=1e-7)
    # indices derived from the flip-flop codes equal the harness's hand-written ones
    m2, s2 = oracle_mod.flipflop_indices(ka["ccrf/seq"], seqlen, 4)
    np.testing.assert_array_equal(m2[:10], move)
    np.testing.assert_array_equal(s2[:12], stay)


def test_c_harness_cat_mod(oracle_mod):
    """c_cat_mod_flipflop.c:589-868: scores -52.354622, -195.435257."""
    ka = load_golden("known_answers.npz")
    lp = ka["ccm/logprob"]
    cost, _ = oracle_mod._seq_call(
        oracle_mod.lib(), "oracle_", lp, ka["ccm/move"], ka["ccm/stay"],
        ka["ccm/seqlen"], ka["ccm/modmoveidx"], ka["ccm/modmovefact"])
    np.testing.assert_allclose(-cost * lp.shape[0], ka["ccm/score"], rtol=2e-7)


@pytest.mark.skipif(not os.path.exists(
    os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "crf_twostate_test")),
    reason="oracle/_ref not built")
def test_ref_harness_binaries_print_known_answers():
    here = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref")
    out = subprocess.run([os.path.join(here, "crf_twostate_test")],
                         capture_output=True, text=True).stdout
    assert "Forwards scores: -2.378088 -2.378088" in out
    assert "Backwards scores: -2.378088 -2.378088" in out
    out = subprocess.run([os.path.join(here, "cat_mod_test")],
                         capture_output=True, text=True).stdout
    assert "Forwards scores: -52.354622 -195.435257" in out


def test_decodeutil_known_logpartition(oracle_mod):
    """test/unit/test_decodeutil.py:16-62: 27.16876983642578."""
    ka = load_golden("known_answers.npz")
    w = ka["decodeutil/weights"][:, None, :]
    # 27.16876983642578 is the partition function when paths may START in any
    # state (decodeutil.forward(init=None), test_decodeutil.py:41-49); with the
    # flip-only start of log_partition_flipflop the same weights give
    # tensor_score (test_decodeutil.py:19-20).  A float64 numpy recursion over
    # the same transition layout reproduces both, and the oracle equals the latter.

    def numpy_logz(w2, init):
        f = np.array(init, dtype=np.float64)
        for row in w2.astype(np.float64):
            m = row[:32].reshape(4, 8) + f[None, :]
            flip = np.log(np.exp(m).sum(axis=1))
            flop = np.logaddexp(f[:4] + row[32:36], f[4:] + row[36:40])
            f = np.concatenate([flip, flop])
        return np.log(np.exp(f).sum())

    w2 = ka["decodeutil/weights"]
    assert abs(numpy_logz(w2, [0] * 8) - float(ka["decodeutil/expt_score"])) < 1e-5
    flip_start = numpy_logz(w2, [0] * 4 + [-50000.0] * 4)
    assert abs(flip_start - float(ka["decodeutil/tensor_score"])) < 1e-5
    lz = oracle_mod.flipflop_logz(w)
    assert abs(float(lz[0]) - float(ka["decodeutil/tensor_score"])) < 1e-5
    lz2, _ = oracle_mod.flipflop_logz_grad(w)
    assert abs(float(lz2[0]) - float(ka["decodeutil/tensor_score"])) < 1e-5


def test_decode_expected_path(oracle_mod):
    """test/unit/test_decode.py:20-54: path [1,0,2,2,1,1,0,0]."""
    ka = load_golden("known_answers.npz")
    fwd, tb, path = oracle_mod.flipflop_viterbi(ka["decode/scores"])
    np.testing.assert_array_equal(path[:, 0], ka["decode/expected_path"])
    np.testing.assert_array_equal(path, ka["decode/path"])
    np.testing.assert_array_equal(tb, ka["decode/tb"])
    np.testing.assert_array_equal(fwd, ka["decode/fwd"])
    _, trans = oracle_mod.flipflop_logz_grad(ka["decode/scores"])
    np.testing.assert_allclose(trans, ka["decode/trans"], atol=2e-6)


def test_viterbi_tie_rule(oracle_mod):
    """All-zero scores: lowest index wins; flop ties go to the flip source."""
    ka = load_golden("known_answers.npz")
    fwd, tb, path = oracle_mod.flipflop_viterbi(np.zeros((5, 2, 40), dtype=np.float32))
    np.testing.assert_array_equal(tb, ka["ties/tb"])
    np.testing.assert_array_equal(path, ka["ties/path"])
    np.testing.assert_array_equal(fwd, ka["ties/fwd"])
    np.testing.assert_array_equal(tb[0, 0], [0, 0, 0, 0, 0, 1, 2, 3])


def test_ctc_loss_probabilities(oracle_mod):
    """test/unit/test_ctc_loss.py:80-103: logZ == 0, P(015)=P(237)=1/2, P(510)=0."""
    ka = load_golden("known_answers.npz")
    outputs = ka["ctcloss/outputs"]
    assert abs(float(oracle_mod.flipflop_logz(outputs)[0])) < 1e-6
    for name in ("015", "237", "510"):
        loss, grad = oracle_mod.crf_flipflop_loss(
            outputs, ka["ctcloss/%s_seq" % name], np.array([3]), 1.0)
        prob = float(np.exp(-loss[0] * outputs.shape[0]))
        assert abs(prob - float(ka["ctcloss/%s_prob" % name])) < 1e-7
        np.testing.assert_allclose(loss, ka["ctcloss/%s_loss" % name], rtol=1e-6)
        np.testing.assert_allclose(grad, ka["ctcloss/%s_grad" % name], atol=1e-6)


# --------------------------------------------------------------------------
# golden fixtures produced by the imported reference Python
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(cases.CRF_SMALL))
def test_crf_small_vs_reference_python(oracle_mod, name):
    gold = load_golden("crf_small.npz")
    spec = cases.CRF_SMALL[name]
    inp = cases.crf_inputs(spec)
    loss, grad = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"],
                                              spec["sharp"])
    np.testing.assert_allclose(loss, gold[name + "/loss"], rtol=1e-6, atol=1e-6)
    _check_grad(gold, name + "/grad", grad, 1e-5)
    loss2, _ = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"],
                                            spec["sharp"], want_grad=False)
    np.testing.assert_allclose(loss2, gold[name + "/loss_nograd"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", list(cases.CATMOD_SMALL))
def test_catmod_small_vs_reference_python(oracle_mod, name):
    gold = load_golden("catmod_small.npz")
    spec = cases.CATMOD_SMALL[name]
    inp = cases.crf_inputs(spec, cases.NMODS)
    loss, grad = oracle_mod.cat_mod_flipflop_loss(
        inp["scores"], inp["seqs"], inp["seqlens"], inp["mod_cats"],
        inp["can_mods_offsets"], inp["mod_cat_weights"], spec["sharp"])
    np.testing.assert_allclose(loss, gold[name + "/loss"], rtol=1e-6, atol=1e-6)
    _check_grad(gold, name + "/grad", grad, 2e-5)


@pytest.mark.parametrize("name", list(cases.LOGZ_SMALL))
def test_logz_viterbi_small_vs_reference_python(oracle_mod, name):
    gold = load_golden("logz_small.npz")
    sc = cases.logz_inputs(cases.LOGZ_SMALL[name])
    lz = oracle_mod.flipflop_logz(sc)
    np.testing.assert_allclose(lz, gold[name + "/logz"], rtol=2e-6)
    lz2, grad = oracle_mod.flipflop_logz_grad(sc)
    np.testing.assert_allclose(lz2, gold[name + "/logz"], rtol=2e-6)
    _check_grad(gold, name + "/grad", grad, 1e-5)
    fwd, tb, path = oracle_mod.flipflop_viterbi(sc)
    np.testing.assert_array_equal(path, gold[name + "/path"])
    np.testing.assert_array_equal(fwd[-1], gold[name + "/fwd_last"])
    if name + "/tb" in gold.files:
        np.testing.assert_array_equal(tb, gold[name + "/tb"])
        np.testing.assert_array_equal(fwd, gold[name + "/fwd"])
        np.testing.assert_allclose(grad, gold[name + "/trans"], atol=1e-5)
    else:
        np.testing.assert_array_equal(tb.sum(axis=(0, 2)), gold[name + "/tb_sum"])


# --------------------------------------------------------------------------
# restatement == genuine reference C (oracle/_ref), same inputs
# --------------------------------------------------------------------------
@pytest.mark.parametrize("T,N,seed", [(60, 5, 3), (257, 9, 4)])
def test_restatement_equals_reference_c(oracle_mod, T, N, seed):
    if not oracle_mod.ref_available():
        pytest.skip("oracle/_ref not built")
    from taiyaki_amd import synth
    inp = synth.crf_case(T, N, seed)
    a, ga = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.3)
    b, gb = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.3,
                                         use_ref=True)
    np.testing.assert_allclose(a, b, rtol=1e-6)
    np.testing.assert_allclose(ga, gb, atol=5e-6)
    inp = synth.crf_case(T, N, seed, nmods_per_base=cases.NMODS)
    a, ga = oracle_mod.cat_mod_flipflop_loss(
        inp["scores"], inp["seqs"], inp["seqlens"], inp["mod_cats"],
        inp["can_mods_offsets"], inp["mod_cat_weights"], 1.0)
    b, gb = oracle_mod.cat_mod_flipflop_loss(
        inp["scores"], inp["seqs"], inp["seqlens"], inp["mod_cats"],
        inp["can_mods_offsets"], inp["mod_cat_weights"], 1.0, use_ref=True)
    np.testing.assert_allclose(a, b, rtol=1e-6)
    np.testing.assert_allclose(ga, gb, atol=2e-5)


def test_float64_witness_agrees_with_the_fp32_restatement_and_the_reference(oracle_mod):
    """oracle_seq_grad_f64 is the same recursion as seq_grad / c_crf_flipflop.c:434-516 with double
    intermediates: on small cases the fp32 results (restatement AND genuine reference) lie within
    fp32 rounding of it -- plain and cat-mod, sharpened, an empty read --, and the distance grows
    with T as rounding noise does (which is why long-T parity is judged against the witness)."""
    from taiyaki_amd import synth
    for T, seqlens, mods, sharp in ((7, [6, 2], None, 1.0), (50, [25, 40, 7, 12], None, 2.5),
                                    (50, [30, 17, 0], None, 1.0), (60, [25, 40, 7], (1, 1, 0, 0), 1.0),
                                    (60, [25, 40, 7], (1, 1, 0, 0), 2.5), (200, [90, 150, 30, 180], None, 1.0)):
        inp = synth.crf_case(T, len(seqlens), 91 + T, seqlens=np.array(seqlens, dtype=np.int32), nmods_per_base=mods)
        margs = (inp["mod_cats"], inp["can_mods_offsets"], inp["mod_cat_weights"]) if mods else ()
        wl, wg = oracle_mod.crf_flipflop_loss_f64(inp["scores"], inp["seqs"], inp["seqlens"], sharp, *margs)
        assert wl.dtype == np.float64 and wg.dtype == np.float64
        for use_ref in ([False, True] if oracle_mod.ref_available() else [False]):
            if mods:
                ol, og = oracle_mod.cat_mod_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], *margs, sharp,
                                                          use_ref=use_ref)
            else:
                ol, og = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], sharp, use_ref=use_ref)
            np.testing.assert_allclose(ol, wl, rtol=2e-6, atol=1e-7)
            assert np.abs(og - wg).max() * T < (3e-4 if mods else 3e-5), (T, mods, sharp, use_ref)
    # the reference's own noise at a long T: larger than at T = 200, far below 1 % (a handful of reads)
    T = 2000
    inp = synth.crf_case(T, 3, 7, seqlens=np.array([900, 1500, 400], dtype=np.int32))
    wl, wg = oracle_mod.crf_flipflop_loss_f64(inp["scores"], inp["seqs"], inp["seqlens"], 1.0)
    ol, og = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0)
    noise = np.abs(og - wg).max() * T
    assert 2e-5 < noise < 5e-3, noise


def test_empty_read_inside_a_batch_is_scored_apart(oracle_mod):
    """The reference gives a read AFTER an empty read that is not the batch's last a neighbour's move index
    (ctc.pyx:127-129 emits no move for the empty read, c_crf_flipflop.c:479 counts -1): shown here on the
    genuine reference when it is built -- at least one later read differs from its value alone --, while the
    oracle's operator-level functions (what the HIP path is held to) give every read the value it has in a
    batch of its own, the empty read cost 0 and zero gradient rows.  Trailing empty reads take the plain route."""
    from taiyaki_amd import synth
    T = 40
    seqlens = np.array([12, 0, 20, 9, 0], dtype=np.int32)
    inp = synth.crf_case(T, len(seqlens), 5, seqlens=seqlens)
    off = np.concatenate([[0], np.cumsum(seqlens)])
    solo = np.zeros(len(seqlens), dtype=np.float32)
    solo_g = np.zeros_like(inp["scores"])
    for n in np.nonzero(seqlens)[0]:
        one, one_g = oracle_mod.crf_flipflop_loss(np.ascontiguousarray(inp["scores"][:, n:n + 1]), inp["seqs"][off[n]:off[n + 1]],
                                         seqlens[n:n + 1], 1.0)
        solo[n], solo_g[:, n] = one[0], one_g[:, 0]
    for use_ref in ([False, True] if oracle_mod.ref_available() else [False]):
        loss, grad = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], seqlens, 1.0, use_ref=use_ref)
        np.testing.assert_array_equal(loss, solo)
        np.testing.assert_allclose(grad, solo_g, rtol=1e-5, atol=1e-8)       # (vector-width remainders round differently)
    wl, _ = oracle_mod.crf_flipflop_loss_f64(inp["scores"], inp["seqs"], seqlens, 1.0)
    np.testing.assert_allclose(wl, solo, rtol=2e-6, atol=1e-7)
    if oracle_mod.ref_available():
        # the raw reference call on the same batch: reads 2 and 3 are not what they are alone
        from oracle import _seq_call, flipflop_indices, ref
        nbase = 4
        move = np.concatenate([flipflop_indices(inp["seqs"][off[n]:off[n + 1]], seqlens[n:n + 1], nbase)[0][:max(seqlens[n] - 1, 0)]
                               for n in range(len(seqlens))] + [np.zeros(2, dtype=np.uintp)])
        stay = flipflop_indices(inp["seqs"], seqlens, nbase)[1]
        raw, _ = _seq_call(ref(), "", inp["scores"], move, stay, seqlens)
        assert raw[0] == solo[0] and (raw[2] != solo[2] or raw[3] != solo[3])
