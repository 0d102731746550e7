"""Drop-in for the flip-flop entry points of ``taiyaki.decode`` (decode.py:15-115)."""
import torch

from taiyaki_amd import _lib, flipflopfings


def flipflop_viterbi(scores, _never_use_cupy=False):
    """decode.py:15-39.  Returns (fwd (T+1,N,2nb) f32, traceback (T,N,2nb) int64,
    path (T+1,N) int64), bit-identical to the reference's torch path
    (decode.py:75-115: first-index tie rule)."""
    del _never_use_cupy
    _lib.require_gpu(scores, "flipflop_viterbi")
    L = _lib.lib()
    sc = scores.detach().float().contiguous()
    T, N, S = sc.shape
    nbase = flipflopfings.nbase_flipflop(S)
    dev = sc.device
    with torch.cuda.device(dev):
        fwd = torch.empty(T + 1, N, 2 * nbase, dtype=torch.float32, device=dev)
        tb = torch.empty(T, N, 2 * nbase, dtype=torch.int64, device=dev)
        path = torch.empty(T + 1, N, dtype=torch.int64, device=dev)
        wsb = L.tk_flipflop_viterbi_workspace_bytes(T, N, nbase)
        ws = _lib.workspace(wsb, dev, "viterbi")         # one scratch buffer per (device, stream), like ctc / layers
        rc = L.tk_flipflop_viterbi_dev(_lib.ptr(sc), T, N, nbase, _lib.ptr(fwd), _lib.ptr(tb),
                                       _lib.ptr(path), _lib.ptr(ws), wsb, _lib.stream_ptr())
        _lib.check(rc, "tk_flipflop_viterbi_dev")
    return fwd, tb, path


def flipflop_viterbi_path(scores):
    """Path-only Viterbi: what bin/basecall.py:222 keeps of `flipflop_viterbi` (the forward
    scores and the int64 traceback tensor, five times the size of the input, are not
    written)."""
    _lib.require_gpu(scores, "flipflop_viterbi_path")
    L = _lib.lib()
    sc = scores.detach().float().contiguous()
    T, N, S = sc.shape
    nbase = flipflopfings.nbase_flipflop(S)
    dev = sc.device
    with torch.cuda.device(dev):
        path = torch.empty(T + 1, N, dtype=torch.int64, device=dev)
        wsb = L.tk_flipflop_viterbi_workspace_bytes(T, N, nbase)
        ws = _lib.workspace(wsb, dev, "viterbi")
        rc = L.tk_flipflop_viterbi_dev(_lib.ptr(sc), T, N, nbase, None, None, _lib.ptr(path),
                                       _lib.ptr(ws), wsb, _lib.stream_ptr())
        _lib.check(rc, "tk_flipflop_viterbi_dev")
    return path


def flipflop_make_trans(scores, _never_use_cupy=False):
    """decode.py:42-72: posterior transition probabilities (not logs) =
    d logZ / d scores; always detached, like the reference."""
    del _never_use_cupy
    from taiyaki_amd.layers import _logz_launch
    _, trans = _logz_launch(scores, True)
    return trans
