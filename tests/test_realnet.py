"""A TRAINED network's scores on REAL reads (tests/golden/realnet.npz, made by make_golden_realnet.py from the
reference's shipped mGru r9 checkpoint and its seven mapped reads): the genuine reference's answers on them
against the oracle (CPU) and against the HIP kernels (-m gpu).

Every other fixture holds iid U(-5, 5) or synthetic "confident" scores; these are 5 tanh outputs of a network
that has learnt its reads -- saturated at +-4.99 on the called path, the rest of each row far below: the
distribution the dispatch rules of the linear CRF path (block length, frame slope, gate) must hold on.
Two cases: `real` (32 chunks of 2000 samples, T = 500, L = 0.33 .. 0.50 T) and `fast` (16 chunks of 3400
samples resampled to 2000: L = 0.62 .. 0.81 T, the narrow bands of fast reads).

Tolerances: loss / logZ <= 1e-5 relative (north_star: 1e-4), gradient samples on the posterior scale <= 2e-4,
Viterbi paths / traceback / remap paths bit for bit, and -- GPU -- `ctc.last_gate_count() == 0`: every read
answered by the linear kernels, none by the log-domain redo.
"""
import numpy as np
import pytest

from tests import parity
from tests.conftest import load_golden
from tests.golden import cases

CASES = ("real", "fast")


def _inputs(gold, tag):
    return dict(scores=gold[tag + "/scores"], seqs=gold[tag + "/seqs"].astype(np.int64),
                seqlens=gold[tag + "/seqlens"].astype(np.int64))


def _check_grad(gold, prefix, grad, T, atol_scaled=2e-4, scaled=None):
    """Against the genuine reference's checksums.  `scaled`: a cat-mod input whose gradient `grad` was handed
    over on the posterior scale (modification columns / their weight) -- only the samples are compared then."""
    cs = cases.grad_checksums(grad)
    np.testing.assert_array_equal(cs["sample_idx"], gold[prefix + "_sample_idx"])
    want = gold[prefix + "_sample"]
    if scaled is None:
        np.testing.assert_allclose(cs["sum"], gold[prefix + "_sum"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(cs["sumsq"], gold[prefix + "_sumsq"], rtol=2e-3, atol=1e-9)
    else:
        want = want * parity.posterior_scale(scaled)[cs["sample_idx"] % grad.shape[2]]
    assert np.abs(cs["sample"] - want).max() * T < atol_scaled, np.abs(cs["sample"] - want).max() * T


# ------------------------------------------------------------------ CPU: the oracle on real scores ----
@pytest.mark.parametrize("tag", CASES)
def test_fixture_is_what_it_says(tag):
    gold = load_golden("realnet.npz")
    sc, seqs, lens = gold[tag + "/scores"], gold[tag + "/seqs"], gold[tag + "/seqlens"]
    T, N, S = sc.shape
    assert (T, S) == (500, 40) and N == len(lens) and sc.dtype == np.float32
    assert np.abs(sc).max() < 5.0 and np.abs(sc).max() > 4.9          # 5 tanh, saturated somewhere
    assert seqs.shape == (lens.sum(),) and seqs.min() >= 0 and seqs.max() < 8
    from oracle import flipflop_code
    assert np.array_equal(np.concatenate([flipflop_code(b, 4) for b in np.split(gold[tag + "/bases"], np.cumsum(lens)[:-1])]),
                          seqs)
    frac = lens / T
    assert (0.3 < frac.min() and frac.max() < 0.55) if tag == "real" else (0.6 < frac.min() and frac.max() < 0.85)
    # a trained network calls its reads: the Viterbi path moves about as often as the read has bases (the shipped
    # network is a small remapping model: a few chunks it calls badly -- real data --, the resampled ones worse)
    moves = (np.diff(gold[tag + "/vit_path"].astype(int), axis=0) != 0).sum(axis=0)
    assert np.median(np.abs(moves - lens) / lens) < (0.12 if tag == "real" else 0.3), (moves, lens)


@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("sharp", [1.0, 2.0])
def test_oracle_crf_on_real_scores(oracle_mod, tag, sharp):
    gold = load_golden("realnet.npz")
    inp = _inputs(gold, tag)
    loss, grad = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], sharp)
    k = "%s/crf_s%d" % (tag, int(sharp))
    assert parity.rel_err(loss, gold[k + "_loss"]) < 1e-5
    _check_grad(gold, k + "_grad", grad, inp["scores"].shape[0])


@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("sharp", [1.0, 2.0])
def test_oracle_catmod_on_real_scores(oracle_mod, tag, sharp):
    gold = load_golden("realnet.npz")
    inp = cases.realnet_catmod_inputs(gold, tag)
    loss, grad = parity.oracle_crf(oracle_mod, inp, sharp)
    k = "%s/catmod_s%d" % (tag, int(sharp))
    assert parity.rel_err(loss, gold[k + "_loss"]) < 1e-5
    _check_grad(gold, k + "_grad", grad * parity.posterior_scale(inp), inp["scores"].shape[0], scaled=inp)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_logz_viterbi_on_real_scores(oracle_mod, tag):
    gold = load_golden("realnet.npz")
    sc = gold[tag + "/scores"]
    lz, lgrad = oracle_mod.flipflop_logz_grad(sc)
    assert parity.rel_err(lz, gold[tag + "/logz"]) < 1e-5
    cs = cases.grad_checksums(lgrad)
    np.testing.assert_allclose(cs["sample"], gold[tag + "/logz_grad_sample"], atol=2e-5)
    np.testing.assert_allclose(lgrad.sum(axis=2), gold[tag + "/trans_rowsum"], atol=1e-4)
    fwd, tb, path = oracle_mod.flipflop_viterbi(sc)
    assert np.array_equal(path, gold[tag + "/vit_path"])
    assert np.array_equal(fwd[-1].view(np.uint32), gold[tag + "/vit_fwd_last"].view(np.uint32))
    assert np.array_equal(tb.sum(axis=(0, 2)), gold[tag + "/vit_tb_sum"])


@pytest.mark.parametrize("tag", CASES)
def test_oracle_remap_and_beam_on_real_scores(tag):
    from oracle import beam
    from oracle import remap as orm
    gold = load_golden("realnet.npz")
    lens = gold[tag + "/seqlens"]
    for n in (0, 1):
        lo = int(lens[:n].sum())
        bases = gold[tag + "/bases"][lo:lo + lens[n]]
        score, path = orm.flipflop_remap(gold[tag + "/scores"][:, n, :], bases, 4)
        assert score == float(gold["%s/remap%d_score" % (tag, n)])
        assert np.array_equal(path, gold["%s/remap%d_path" % (tag, n)])
    seq, score = beam.beamsearch(gold[tag + "/scores"][:, 0, :], 0.0, 5, True)
    assert np.array_equal(seq, gold[tag + "/beam0_seq"]) and np.float32(score) == gold[tag + "/beam0_score"]


# ------------------------------------------------------------------ GPU: the HIP kernels on real scores ----
@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("sharp", [1.0, 2.0])
def test_hip_crf_on_real_scores_stays_on_the_linear_path(oracle_mod, gpu_device, tag, sharp):
    """crf_flipflop_loss as bin/train_flipflop.py:161-173 calls it, against the GENUINE reference's numbers and
    the float64 witness; no read disowned."""
    from taiyaki_amd import _lib, ctc
    gold = load_golden("realnet.npz")
    inp = _inputs(gold, tag)
    r = parity.compare_crf(oracle_mod, inp, sharp, gpu_device)
    assert _lib.is_strict() and ctc.last_gate_count() == 0, ctc.last_gate_count()
    assert r["finite"]
    k = "%s/crf_s%d" % (tag, int(sharp))
    assert parity.rel_err(r["loss"], gold[k + "_loss"]) < 1e-5
    assert parity.crf_grad_ok(r), (r["grad_f64_scaled"], r["grad_scaled_abs"], r["ref_noise_scaled"])
    _check_grad(gold, k + "_grad", r["grad"], inp["scores"].shape[0])
    # cost-only call, and labels that live on the device with no length hint (train_abinitio.py:207-210)
    c0, _ = parity.run_crf(inp, sharp, gpu_device, want_grad=False)
    assert parity.rel_err(c0, gold[k + "_loss"]) < 1e-5 and ctc.last_gate_count() == 0
    c1, g1 = parity.run_crf(inp, sharp, gpu_device, seq_on_device=True)
    assert parity.rel_err(c1, gold[k + "_loss"]) < 1e-5 and ctc.last_gate_count() == 0
    _check_grad(gold, k + "_grad", g1, inp["scores"].shape[0])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
@pytest.mark.parametrize("sharp", [1.0, 1.3, 2.0, 2.5])
def test_hip_catmod_on_real_scores_stays_on_the_linear_path(oracle_mod, gpu_device, tag, sharp):
    """cat_mod_flipflop_loss under the reference's --sharpen schedule (bin/train_flipflop.py:161-170 sharpens the
    cat-mod loss too; round 5's probe had 8 of 32 such reads disowned at 2.0).  Against the genuine reference at
    1.0 / 2.0, the oracle + float64 witness everywhere; no read disowned."""
    from taiyaki_amd import ctc
    gold = load_golden("realnet.npz")
    inp = cases.realnet_catmod_inputs(gold, tag)
    r = parity.compare_crf(oracle_mod, inp, sharp, gpu_device)
    assert ctc.last_gate_count() == 0, ctc.last_gate_count()
    assert r["finite"] and r["loss_rel"] < 1e-5, r["loss_rel"]
    assert parity.crf_grad_ok(r), (r["grad_f64_scaled"], r["grad_scaled_abs"], r["ref_noise_scaled"])
    k = "%s/catmod_s%d" % (tag, int(sharp))
    if sharp == int(sharp):                 # (the genuine reference's numbers are held for 1.0 and 2.0)
        assert parity.rel_err(r["loss"], gold[k + "_loss"]) < 1e-5
        _check_grad(gold, k + "_grad", r["grad"] * parity.posterior_scale(inp), inp["scores"].shape[0], scaled=inp)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_logz_viterbi_make_trans_on_real_scores(gpu_device, tag):
    import torch
    from taiyaki_amd import decode
    gold = load_golden("realnet.npz")
    sc = gold[tag + "/scores"]
    lz, lgrad = parity.run_logz(sc, gpu_device)
    assert parity.rel_err(lz, gold[tag + "/logz"]) < 1e-5
    cs = cases.grad_checksums(lgrad)
    np.testing.assert_allclose(cs["sum"], gold[tag + "/logz_grad_sum"], rtol=1e-4)
    np.testing.assert_allclose(cs["sumsq"], gold[tag + "/logz_grad_sumsq"], rtol=1e-3)
    np.testing.assert_allclose(cs["sample"], gold[tag + "/logz_grad_sample"], atol=2e-5)
    fwd, tb, path = parity.run_viterbi(sc, gpu_device)
    assert np.array_equal(path, gold[tag + "/vit_path"])
    assert np.array_equal(fwd[-1].view(np.uint32), gold[tag + "/vit_fwd_last"].view(np.uint32))
    assert np.array_equal(tb.sum(axis=(0, 2)), gold[tag + "/vit_tb_sum"])
    trans = decode.flipflop_make_trans(torch.from_numpy(sc).to(gpu_device)).cpu().numpy()
    np.testing.assert_allclose(trans.sum(axis=2), gold[tag + "/trans_rowsum"], atol=1e-4)
    np.testing.assert_allclose(trans, lgrad, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_lossvector_on_real_scores(gpu_device, tag):
    """calculate_loss's assembly (bin/train_flipflop.py:172-182) through the fused operator."""
    import torch
    from taiyaki_amd import ctc
    gold = load_golden("realnet.npz")
    inp = _inputs(gold, tag)
    x = torch.from_numpy(inp["scores"]).to(gpu_device).requires_grad_()
    lv = ctc.flipflop_loss(x, torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"]), 1.0)
    assert ctc.last_gate_count() == 0
    # the two terms cancel to ~1e-3 of their size: an absolute bound on the per-block loss
    np.testing.assert_allclose(lv.detach().cpu().numpy(), gold[tag + "/lossvector"], atol=5e-5)
    loss, lv2 = ctc.flipflop_mean_loss(x, torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"]), 1.0)
    assert ctc.last_gate_count() == 0 and torch.equal(lv2, lv.detach())
    assert abs(float(loss) - float(gold[tag + "/lossvector"].astype(np.float64).mean())) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_remap_and_beam_on_real_scores(gpu_device, tag):
    from taiyaki_amd import decodeutil
    from taiyaki_amd import flipflop_remap as fr
    gold = load_golden("realnet.npz")
    lens = gold[tag + "/seqlens"]
    for n in (0, 1):
        lo = int(lens[:n].sum())
        seq = "".join("ACGT"[b] for b in gold[tag + "/bases"][lo:lo + lens[n]])
        score, path = fr.flipflop_remap(gold[tag + "/scores"][:, n, :], seq, alphabet="ACGT")
        assert score == float(gold["%s/remap%d_score" % (tag, n)])
        assert np.array_equal(path, gold["%s/remap%d_path" % (tag, n)])
        bseq, bscore = decodeutil.beamsearch(np.ascontiguousarray(gold[tag + "/scores"][:, n, :]), 0.0, 5, True)
        assert np.array_equal(bseq, gold["%s/beam%d_seq" % (tag, n)])
        assert abs(bscore - float(gold["%s/beam%d_score" % (tag, n)])) <= 2e-6 * abs(bscore)
