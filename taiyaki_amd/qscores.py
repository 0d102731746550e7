"""Drop-in for the basecaller's quality-score helpers (``taiyaki/qscores.py``).

``errprobs_from_trans`` -- the only one that touches the (T, N, S) posterior tensor -- runs as
one gfx950 HIP pass (csrc/qscore_kernels.hip) instead of nbase masked matmuls plus a
normalise / gather chain.  The string helpers are small host-side restatements with the
reference's names and argument meaning.
"""
import numpy as np
import torch

from taiyaki_amd import _lib, flipflopfings


def qchar_from_qscore(score, zerochar=33):
    """qscores.py:10-27: ASCII code = score + zerochar, rounded to nearest."""
    codes = (np.asarray(score) + zerochar + 0.5).astype(np.int8)
    return codes.tobytes().decode("ascii")


def qscore_from_errprob(errprob):
    """qscores.py:30-39: -10 log10(errprob)."""
    return -10.0 * np.log10(errprob)


def qchar_from_errprob(errprob, qscore_scale, qscore_offset):
    """qscores.py:42-55"""
    return qchar_from_qscore(qscore_scale * qscore_from_errprob(errprob) + qscore_offset)


def transitions_into_base(b, nbases, device=None):
    """Columns of the transition-score vector that END in base b (the interface of
    qscores.py:58-85; csrc/qscore_kernels.hip sums exactly these per block).  The vector is the
    (nbases + 1) x 2 nbases table `[to][from]` of layers.py:1253-1274 read row by row: row b holds
    every transition into b's flip state, and the last row (the flops) holds flip b -> flop b and
    the flop b stay in its columns b and b + nbases."""
    table = torch.arange(2 * nbases * (nbases + 1), dtype=torch.long, device=device).view(nbases + 1, 2 * nbases)
    return torch.cat((table[b], table[nbases, [b, b + nbases]]))


def errprobs_from_trans(trans, path):
    """qscores.py:88-142.  trans (nblocks, batch, nstates) posterior transition weights,
    path (nblocks + 1, batch) flip-flop states -> (nblocks + 1, batch) error probabilities,
    -1 in row 0."""
    _lib.require_gpu(trans, "errprobs_from_trans")
    L = _lib.lib()
    tr = trans.detach().float().contiguous()
    if tr.data_ptr() % 16 != 0:
        tr = tr.clone()
    nblk, nbatch, nstate = tr.shape
    nbase = flipflopfings.nbase_flipflop(nstate)
    pth = path.to(device=tr.device, dtype=torch.int64).contiguous()
    if tuple(pth.shape) != (nblk + 1, nbatch):
        raise ValueError("path must have shape (nblocks + 1, batch)")
    with torch.cuda.device(tr.device):
        out = torch.empty((nblk + 1, nbatch), dtype=torch.float32, device=tr.device)
        rc = L.tk_flipflop_errprobs_dev(_lib.ptr(tr), _lib.ptr(pth), nblk, nbatch, nbase,
                                        _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, "tk_flipflop_errprobs_dev")
    return out


def path_errprobs_to_qstring(errprobs, path, qscore_scale, qscore_offset):
    """qscores.py:145-178: quality characters for the emitted bases only (stays and the
    source state of the first transition are skipped)."""
    picked = errprobs[1:][path[1:] != path[:-1]]
    if isinstance(picked, torch.Tensor):
        picked = picked.detach().cpu().numpy()
    return qchar_from_errprob(picked, qscore_scale, qscore_offset)
