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
