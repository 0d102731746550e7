"""Kernel A's band-mode schedule (csrc/crf_band.hip) restated in numpy
(tests/helpers/crf_skew_model.py) against the oracle, on the CPU: skewed chunk/block
phases, integer per-chunk log2 offsets, the live-band windows, the sorted-instance
posterior.  The HIP kernel mirrors the model name for name; the -m gpu parity tests
check the kernel itself."""
import numpy as np
import pytest

from tests.helpers import crf_skew_model as model


@pytest.mark.parametrize("T,Ls,PW", [
    (20, [9, 1, 21, 20, 2], 4),         # L = 1, L = T, L = T + 1 (every block moves)
    (37, [12, 30, 38, 5], 4),
    (64, [33, 50, 7], 8),
    (50, [25, 26], 64),                 # one chunk
    (19, [20, 3], 2),                   # T not a multiple of the time block
    (45, [23], 16),
])
def test_band_schedule_model_matches_oracle(oracle_mod, T, Ls, PW):
    from taiyaki_amd import synth
    inp = synth.crf_case(T, len(Ls), 3 + T, seqlens=np.array(Ls, dtype=np.int32))
    oloss, ograd = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0)
    mv, stv = oracle_mod.flipflop_indices(inp["seqs"], inp["seqlens"], 4)
    off = np.concatenate([[0], np.cumsum(Ls)])
    for n, L in enumerate(Ls):
        st = stv[off[n]:off[n] + L].astype(int)
        mo = mv[off[n] - n:off[n] - n + L - 1].astype(int)
        cost, grad, dbg = model.crf_model(inp["scores"][:, n], st, mo, L, PW)
        assert abs(cost - oloss[n]) <= 2e-6 * abs(oloss[n]), (L, cost, oloss[n])
        assert np.abs(grad - ograd[:, n]).max() < 2e-6
        # both sweeps agree on the score (c_crf_flipflop.c:482-491 averages them)
        assert abs(dbg[0] - dbg[1]) < 1e-3


def test_band_windows_cover_exactly_the_cells_on_complete_paths():
    """Every cell on a complete path lies in a live (chunk, block); the posterior pass only
    reads stored columns (the model would propagate NaN otherwise)."""
    for T, L, PW, KB in ((40, 17, 4, 8), (33, 34, 8, 8), (100, 3, 2, 8), (64, 64, 16, 8)):
        win = model.windows(L, T, PW, KB)
        for t in range(T):
            for p in range(L):
                if p <= t and L - 1 - p <= T - t:      # column t, on a complete path
                    j0, j1 = win[p // PW]
                    assert j0 <= t // KB <= j1 or t == T, (T, L, t, p)


# ---------------------------------------------------------------------------------------------
# round 3: the LINEAR-domain arithmetic of the band kernel (tests/helpers/crf_linear_model.py)
# ---------------------------------------------------------------------------------------------
def _reads(oracle_mod, T, Ls, seed=None):
    from taiyaki_amd import synth
    inp = synth.crf_case(T, len(Ls), (3 + T) if seed is None else seed, seqlens=np.array(Ls, dtype=np.int32))
    mv, stv = oracle_mod.flipflop_indices(inp["seqs"], inp["seqlens"], 4)
    off = np.concatenate([[0], np.cumsum(Ls)])
    for n, L in enumerate(Ls):
        yield n, L, inp, stv[off[n]:off[n] + L].astype(int), mv[off[n] - n:off[n] - n + L - 1].astype(int)


@pytest.mark.parametrize("T,Ls,PW", [
    (20, [9, 1, 21, 20, 2], 4),         # L = 1, L = T, L = T + 1 (every block moves)
    (37, [12, 30, 38, 5], 4),
    (64, [33, 50, 7], 8),
    (50, [25, 26], 64),                 # one chunk
    (19, [20, 3], 2),                   # T not a multiple of the time block
    (200, [90, 150, 30, 180], 64),      # cliffs of 2^60+ between neighbouring cells near the diagonal front
    (400, [200, 266, 350], 64),
])
def test_linear_band_model_matches_oracle(oracle_mod, T, Ls, PW):
    from tests.helpers import crf_linear_model as lin
    oloss = ograd = None
    for n, L, inp, st, mo in _reads(oracle_mod, T, Ls):
        if oloss is None:
            oloss, ograd = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0)
        cost, grad, info = lin.crf_model(inp["scores"][:, n], st, mo, L, PW)
        assert not info["bad"], (T, L, info)
        assert abs(cost - oloss[n]) <= 2e-6 * abs(oloss[n]), (L, cost, oloss[n])
        assert np.abs(grad - ograd[:, n]).max() < 2e-6
        # every row's posterior total is the partition function (the kernel's mass-loss detector)
        assert info["rowz_dev"] < 1e-4
        assert abs(info["scoreF"] - info["scoreB"]) < 1e-4


def test_linear_band_model_disowns_what_it_cannot_represent(oracle_mod):
    """Bands a few cells wide lose their (tiny) front cells to the frames' flush, sharpened scores
    overflow inside a block: the model -- like the kernel -- must SAY so (non-finite score, or a
    row whose posterior total is not the partition function), never return a wrong number."""
    from tests.helpers import crf_linear_model as lin
    flagged = 0
    for T, Ls, sharp in ((200, [201, 199, 195], 1.0), (400, [390, 401], 1.0), (200, [100, 150], 2.5)):
        oloss = None
        for n, L, inp, st, mo in _reads(oracle_mod, T, Ls):
            if oloss is None:
                oloss, ograd = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], sharp)
            cost, grad, info = lin.crf_model(inp["scores"][:, n], st, mo, L, 64, sharp=sharp)
            disowned = info["bad"] or not (info["rowz_dev"] < 1e-3)
            flagged += disowned
            if not disowned:
                assert abs(cost - oloss[n]) <= 2e-6 * abs(oloss[n])
                assert np.abs(grad - ograd[:, n]).max() < 2e-6
    assert flagged >= 3


def test_linear_band_model_catmod(oracle_mod):
    from taiyaki_amd import synth
    from tests.helpers import crf_linear_model as lin
    T, Ls = 60, [25, 40, 7]
    inp = synth.crf_case(T, len(Ls), 11, seqlens=np.array(Ls, dtype=np.int32), nmods_per_base=(1, 1, 0, 0))
    wts = (inp["mod_cat_weights"] * 0.125).astype(np.float32)
    oloss, ograd = oracle_mod.cat_mod_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], inp["mod_cats"],
                                                    inp["can_mods_offsets"], wts, 1.0)
    mv, stv = oracle_mod.flipflop_indices(inp["seqs"], inp["seqlens"], 4)
    mdv, mfv = oracle_mod.cat_mod_indices(inp["seqs"], inp["seqlens"], inp["mod_cats"], inp["can_mods_offsets"], wts, 4)
    off = np.concatenate([[0], np.cumsum(Ls)])
    for n, L in enumerate(Ls):
        sl = slice(off[n] - n, off[n] - n + L - 1)
        cost, grad, info = lin.crf_model(inp["scores"][:, n], stv[off[n]:off[n] + L].astype(int), mv[sl].astype(int),
                                         L, 16, mod=mdv[sl].astype(int), modfact=mfv[sl].astype(np.float32))
        assert not info["bad"]
        assert abs(cost - oloss[n]) <= 2e-6 * abs(oloss[n])
        assert np.abs(grad - ograd[:, n]).max() < 2e-6


def test_sweep_liveness_closed_form_equals_the_windows():
    """The gradient pass asks, per 64-cell chunk, whether the SWEEP chunk holding a cell ran the wave's time
    block; crf_band.hip answers with band_window solved for the block once per wave (`sweep_live`: chunk
    start a is live in block jb iff a <= t0 + BK and a + PWS - 1 >= t0 - (T - L + 1), every block for reads
    without a complete path).  Here: that closed form against the schedule model's windows, exhaustively
    over small shapes (the kernel's bit-for-bit regression runs cover the real ones)."""
    from tests.helpers import crf_skew_model as sk
    KB = 8
    for T in (1, 7, 8, 9, 23, 40, 64, 65, 100):
        for L in list(range(1, min(T + 4, 70))) + [T + 1, T + 2, 2 * T + 5]:
            for PWS in (4, 8, 16):
                win = sk.windows(L, T, PWS, KB)
                NB = (T + KB - 1) // KB
                notrim = L > T + 1
                for jb in range(NB):
                    t0 = jb * KB
                    hi, lo = t0 + KB, t0 - (T - L + 1) - PWS + 1
                    for p in range(-1, len(win) * PWS + PWS):
                        a = (p // PWS) * PWS
                        closed = p >= 0 and a < L and (notrim or (a <= hi and a >= lo))
                        w = p // PWS
                        ref = 0 <= w < len(win) and win[w][0] <= jb <= win[w][1]
                        assert closed == ref, (T, L, PWS, jb, p)


def test_narrow_bands_need_steeper_frames(oracle_mod):
    """Round 5 (tools/crf_gate_band_probe.py, LABNOTES R5.19-R5.20): a narrow band carries its mass along the fronts of the
    two lattices, where values fall by 15-30 bits per cell -- faster than frames of slope KLIP = 6 follow; the front's
    cells are flushed and the sweeps disagree (the kernel disowns such a read).  With 8-step blocks, bias 3 and slope
    11 (what crf_band_pick_block gives a batch with narrow bands: 8 x (7.2 - 3 + 11) = 121.6 bits of growth) the same
    reads come out to 1e-5 bit.  The model loses the mass the kernel loses, at the same cell."""
    from taiyaki_amd import synth
    from tests.helpers import crf_linear_model as lin
    T = 800
    cases = (((1, 1, 0, 0), [600, 560, 620, 400], 2), (None, [600, 560, 620, 400, 680, 720], 5))
    try:
        for mods, Ls, n in cases:
            inp = synth.crf_case(T, len(Ls), 3, seqlens=np.array(Ls, dtype=np.int32), nmods_per_base=mods)
            mv, stv = oracle_mod.flipflop_indices(inp["seqs"], inp["seqlens"], 4)
            mdv = mfv = None
            if mods:
                synth.normalise_mod_columns(inp)
                wts = inp["mod_cat_weights"].astype(np.float32)
                mdv, mfv = oracle_mod.cat_mod_indices(inp["seqs"], inp["seqlens"], inp["mod_cats"], inp["can_mods_offsets"], wts, 4)
                oloss, _ = oracle_mod.cat_mod_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], inp["mod_cats"],
                                                            inp["can_mods_offsets"], wts, 1.0)
            else:
                oloss, _ = oracle_mod.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0)
            off = np.concatenate([[0], np.cumsum(Ls)])
            L = Ls[n]
            sl = slice(off[n] - n, off[n] - n + L - 1)
            args = (inp["scores"][:, n], stv[off[n]:off[n] + L].astype(int), mv[sl].astype(int), L, 64)
            kw = dict(mod=mdv[sl].astype(int), modfact=mfv[sl].astype(np.float32)) if mods else {}
            lin.KLIP, lin.WBIAS = 6, 0
            _, _, info = lin.crf_model(*args, KB=8, NORM=8, **kw)
            assert abs(info["scoreF"] - info["scoreB"]) > 10.0, (mods, info)          # bits of mass gone
            lin.KLIP, lin.WBIAS = 11, 3
            cost, _, info = lin.crf_model(*args, KB=8, NORM=8, **kw)
            assert not info["bad"] and abs(info["scoreF"] - info["scoreB"]) < 1e-4, (mods, info)
            assert abs(cost - oloss[n]) <= 1e-5 * abs(oloss[n]), (mods, cost, oloss[n])     # (the fp32 reference is the noisier side at T = 800)
    finally:
        lin.KLIP, lin.WBIAS = 6, 0
