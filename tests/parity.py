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


def run_crf(inp, sharp, dev, want_grad=True, seq_on_device=False):
    from taiyaki_amd import ctc
    x = _t(inp["scores"], dev).requires_grad_(want_grad)
    seqs = _t(inp["seqs"], dev if seq_on_device else "cpu")
    seqlens = _t(inp["seqlens"], dev if seq_on_device else "cpu")
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


def compare_crf(oracle, inp, sharp, dev, **kw):
    loss, grad = run_crf(inp, sharp, dev, **kw)
    oloss, ograd = oracle_crf(oracle, inp, sharp)
    T = inp["scores"].shape[0]
    ncan = 40 if inp["scores"].shape[2] >= 40 else inp["scores"].shape[2]
    live = np.asarray(inp["seqlens"]) > 0
    rowsum = grad[:, live, :ncan].sum(axis=2) * T if live.any() else np.zeros(1)
    return dict(loss_rel=rel_err(loss, oloss), loss_abs=abs_err(loss, oloss),
                grad_abs=abs_err(grad, ograd), grad_scaled_abs=abs_err(grad * T, ograd * T),
                rowsum_dev=float(np.max(np.abs(rowsum + 1.0))),
                finite=bool(np.isfinite(loss).all() and np.isfinite(grad).all()),
                loss=loss, grad=grad, oloss=oloss, ograd=ograd)


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
    return cases.crf_inputs(dict(T=spec["T"], N=spec["N"], seed=spec["seed"]), spec["mods"])
