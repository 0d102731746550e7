"""taiyaki_amd -- MI355X (gfx950) native flip-flop CRF training hot path for Taiyaki.

Operator API (same names as the reference):
    taiyaki_amd.ctc.crf_flipflop_loss        <- taiyaki.ctc.crf_flipflop_loss
    taiyaki_amd.ctc.cat_mod_flipflop_loss    <- taiyaki.ctc.cat_mod_flipflop_loss
    taiyaki_amd.layers.flipflop_logpartition <- taiyaki.layers.flipflop_logpartition
    taiyaki_amd.decode.flipflop_viterbi      <- taiyaki.decode.flipflop_viterbi
    taiyaki_amd.decode.flipflop_make_trans   <- taiyaki.decode.flipflop_make_trans
All of them run hand-written HIP kernels behind the C ABI declared in
include/taiyaki_amd_flipflop.h; there is no CPU fallback.
"""
from taiyaki_amd._lib import build, raise_if_nonfinite, set_strict  # noqa: F401

__all__ = ["build", "raise_if_nonfinite", "set_strict"]
