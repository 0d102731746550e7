"""SURVEY 8f.2 -- basecall-side consumers: posterior transition weights, Viterbi path,
per-block error probabilities (qscores.py:88-142), chunk stitching
(basecall_helpers.py:46-94), quality strings (qscores.py:145-178).

CPU: the oracle restatements against fixtures produced by the genuine reference
(tests/golden/make_golden_basecall.py).  GPU: the HIP path against both."""
import os

import numpy as np
import pytest

from tests.golden import cases

HERE = os.path.dirname(os.path.abspath(__file__))


def gold():
    return np.load(os.path.join(HERE, "golden", "basecall_small.npz"))


@pytest.mark.parametrize("name", list(cases.BASECALL_SMALL))
def test_oracle_matches_reference_goldens(oracle_mod, name):
    g, spec = gold(), cases.BASECALL_SMALL[name]
    scores = cases.basecall_scores(spec)
    _, trans = oracle_mod.flipflop_logz_grad(scores)
    _, _, path = oracle_mod.flipflop_viterbi(scores)
    np.testing.assert_array_equal(path, g[name + "/path"])                    # bit-exact
    np.testing.assert_allclose(trans.sum(axis=2), g[name + "/trans_sum"], atol=2e-5)
    err = oracle_mod.errprobs_from_trans(trans, path)
    np.testing.assert_allclose(err, g[name + "/errprobs"], atol=2e-5)
    starts, ends = g[name + "/chunk_starts"], g[name + "/chunk_ends"]
    sp = oracle_mod.stitch_chunks(path, starts, ends, spec["stride"])
    np.testing.assert_array_equal(sp, g[name + "/stitched_path"])
    np.testing.assert_array_equal(
        oracle_mod.stitch_chunks(path, starts, ends, spec["stride"], path_stitching=True),
        g[name + "/stitched_path_ps"])
    np.testing.assert_allclose(oracle_mod.stitch_chunks(err, starts, ends, spec["stride"]),
                               g[name + "/stitched_errprobs"], atol=2e-5)


def test_host_string_helpers_match_reference_goldens():
    """qchar / qstring helpers and stitch_chunks are host code: checked without a GPU on the
    reference's own error probabilities."""
    import torch
    from taiyaki_amd import basecall_helpers, qscores
    g = gold()
    for name, spec in cases.BASECALL_SMALL.items():
        starts, ends = g[name + "/chunk_starts"], g[name + "/chunk_ends"]
        path, err = torch.from_numpy(g[name + "/path"]), torch.from_numpy(g[name + "/errprobs"])
        sp = basecall_helpers.stitch_chunks(path, starts, ends, spec["stride"])
        se = basecall_helpers.stitch_chunks(err, starts, ends, spec["stride"])
        np.testing.assert_array_equal(sp.numpy(), g[name + "/stitched_path"])
        np.testing.assert_array_equal(se.numpy(), g[name + "/stitched_errprobs"])
        np.testing.assert_array_equal(
            basecall_helpers.stitch_chunks(path, starts, ends, spec["stride"], path_stitching=True).numpy(),
            g[name + "/stitched_path_ps"])
        q = qscores.path_errprobs_to_qstring(se, sp.numpy(), 0.9, 0.3)
        assert q == bytes(g[name + "/qstring"]).decode("ascii")
    idx = qscores.transitions_into_base(2, 4).numpy()
    np.testing.assert_array_equal(idx, [16, 17, 18, 19, 20, 21, 22, 23, 34, 38])
    assert qscores.qchar_from_qscore([0, 1.4, 40]) == '!"I'


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cases.BASECALL_SMALL))
def test_hip_basecall_chain_matches_reference(oracle_mod, gpu_device, name):
    """scores -> make_trans -> viterbi -> errprobs -> stitch -> qstring, all on the HIP path."""
    import torch
    from taiyaki_amd import basecall_helpers, decode, qscores
    g, spec = gold(), cases.BASECALL_SMALL[name]
    scores = torch.from_numpy(cases.basecall_scores(spec)).to(gpu_device)
    trans = decode.flipflop_make_trans(scores)
    _, _, path = decode.flipflop_viterbi(scores)
    np.testing.assert_array_equal(path.cpu().numpy(), g[name + "/path"])      # bit-exact
    assert torch.equal(decode.flipflop_viterbi_path(scores), path)            # path-only variant
    err = qscores.errprobs_from_trans(trans, path)
    assert err.shape == (spec["T"] + 1, spec["N"]) and err.device == scores.device
    # HIP kernel vs the oracle on identical inputs, then vs the reference end to end
    np.testing.assert_allclose(err.cpu().numpy(),
                               oracle_mod.errprobs_from_trans(trans.cpu().numpy(), path.cpu().numpy()),
                               atol=1e-6)
    np.testing.assert_allclose(err.cpu().numpy(), g[name + "/errprobs"], atol=2e-5)
    starts, ends = g[name + "/chunk_starts"], g[name + "/chunk_ends"]
    sp = basecall_helpers.stitch_chunks(path, starts, ends, spec["stride"]).cpu().numpy()
    se = basecall_helpers.stitch_chunks(err, starts, ends, spec["stride"])
    np.testing.assert_array_equal(sp, g[name + "/stitched_path"])
    q = qscores.path_errprobs_to_qstring(se, sp, 0.9, 0.3)
    ref = bytes(g[name + "/qstring"]).decode("ascii")
    assert len(q) == len(ref)
    # a quality character is a rounded -10 log10(p): allow one step where p sits on a boundary
    diff = np.abs(np.frombuffer(q.encode(), np.uint8).astype(int) - np.frombuffer(ref.encode(), np.uint8))
    assert diff.max(initial=0) <= 1 and (diff > 0).mean() < 0.01


@pytest.mark.gpu
def test_hip_errprobs_fullsize_properties(gpu_device):
    """At BASELINE size (T=800, N=128): rows are probabilities, row 0 is -1, and the op is
    invariant to permuting reads."""
    import torch
    from taiyaki_amd import decode, qscores, synth
    sc = torch.from_numpy(synth.scores(800, 128, 40, 5)).to(gpu_device)
    trans = decode.flipflop_make_trans(sc)
    _, _, path = decode.flipflop_viterbi(sc)
    err = qscores.errprobs_from_trans(trans, path)
    assert torch.all(err[0] == -1.0)
    assert float(err[1:].min()) >= -1e-6 and float(err[1:].max()) <= 1.0 + 1e-6
    perm = torch.randperm(128, device=gpu_device)
    err2 = qscores.errprobs_from_trans(trans[:, perm].contiguous(), path[:, perm].contiguous())
    assert torch.equal(err2, err[:, perm])
    with pytest.raises(RuntimeError):
        qscores.errprobs_from_trans(trans.cpu(), path.cpu())


# ------------------------------------------------------------------------------------------
# SURVEY 8f.1 -- on-device gradient maxima / clipping (bin/train_flipflop.py:201-212)
# ------------------------------------------------------------------------------------------
def test_rolling_mad_matches_reference_goldens():
    from taiyaki_amd import clipping
    g = gold()
    vals = g["rollingmad/vals"]
    for tag, n_mads in (("m0", 0), ("m15", 1.5)):
        rm = clipping.RollingMAD(5, n_mads=n_mads, window=6)
        for v, want in zip(vals, g["rollingmad/thresh_" + tag]):
            th = rm.update(list(v))
            if np.isnan(want).all():
                assert th is None
            else:
                np.testing.assert_allclose(th, want, rtol=1e-6)
    with pytest.raises(AssertionError):
        clipping.RollingMAD(5).update([1.0, 2.0])


def test_rolling_mad_with_nonfinite_maxima_is_the_median_over_the_raw_window():
    """Round-4 advisor finding: +inf in the window made the threshold NaN ("no clipping") for `window`
    steps, while the reference's np.median over the raw window (maths.py:182-195) stays finite for a single
    +inf and keeps clipping; NaN does poison it.  Compared with the definition evaluated directly on the last
    `window` rows: median + n_mads * 1.4826 * median(|x - median|)."""
    from taiyaki_amd import clipping
    rs = np.random.RandomState(5)
    window, nparams, n_mads = 7, 6, 2.0
    vals = np.abs(rs.standard_normal((40, nparams))).astype(np.float32)
    vals[9, 1] = np.inf                     # one +inf: stays finite
    vals[12, 2] = np.nan                    # one NaN: NaN while it is in the window
    vals[20:24, 3] = np.inf                 # four of seven +inf: centre inf -> NaN like the reference
    vals[30, 4] = np.inf
    vals[31, 4] = np.nan
    rm = clipping.RollingMAD(nparams, n_mads=n_mads, window=window)
    for t in range(len(vals)):
        th = rm.update(vals[t])
        if t + 1 < window:
            assert th is None
            continue
        w = vals[t + 1 - window:t + 1]
        with np.errstate(invalid="ignore"):
            med = np.median(w, axis=0)
            want = med + n_mads * clipping.MAD_SD_FACTOR * np.median(np.abs(w - med[None]), axis=0)
        assert np.array_equal(np.isnan(th), np.isnan(want)), t
        ok = ~np.isnan(want)
        np.testing.assert_allclose(np.asarray(th)[ok], want[ok], rtol=1e-6)
        if 9 <= t < 9 + window:
            assert np.isfinite(th[1])


@pytest.mark.gpu
def test_device_clipper_matches_apply_clipping(gpu_device):
    """Kernel maxima == per-tensor max|grad| (the reference's grad_maxs), clamp == the
    reference's clamp-if-exceeds, thresholds follow the rolling MAD one step behind."""
    import torch
    from taiyaki_amd import clipping, models, parallel
    torch.manual_seed(5)
    net = models.mLstm_flipflop(size=32, stride=5).to(gpu_device)
    arena = parallel.FlatGradArena(net)
    clip = clipping.DeviceClipper(arena, n_mads=0, window=3)
    ref_roll = clipping.RollingMAD(len(arena.params), n_mads=0, window=3)
    thresh = None
    for it in range(6):
        arena.flat.copy_(torch.randn_like(arena.flat) * (1.0 + it))
        before = arena.flat.clone()
        want_maxs = np.array([float(p.grad.abs().max()) for p in arena.params], dtype=np.float32)
        prev = clip.step()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(clip.maxs.cpu().numpy(), want_maxs)
        # reference order: clip with the thresholds known before this step, then update them
        expect = before.clone()
        if thresh is not None:
            off = 0
            for p, th in zip(arena.params, thresh):
                seg = expect[off:off + p.numel()]
                if float(seg.abs().max()) > th:
                    seg.clamp_(min=-float(th), max=float(th))
                off += p.numel()
        assert torch.equal(arena.flat, expect), it
        if it > 0:
            np.testing.assert_array_equal(prev, last_maxs)
        last_maxs = want_maxs
        thresh = ref_roll.update(want_maxs)
    # a NaN gradient surfaces as a NaN maximum, like float(torch.max(...))
    arena.flat[5] = float("nan")
    clip.step()
    torch.cuda.synchronize()
    assert np.isnan(clip.maxs.cpu().numpy()[0])
