"""Drop-in for ``taiyaki.decodeutil`` (taiyaki/decodeutil/decodeutil.pyx): ``beamsearch`` (9-51),
the hash beam search over the flip-flop lattice, and the decoder's ``forward`` / ``backward``
lattice passes (54-108), on the gfx950 kernels of csrc/beam_kernels.hip.

The reference decodes one read per call on the host; here a call takes one read ``(T, S)`` --
same return value ``(sequence, score)`` -- or a batch ``(T, N, S)`` (one wavefront per read, one
launch) and then returns ``(list of sequences, scores)``.
"""
import numpy as np
import torch

from taiyaki_amd import _lib, flipflopfings


def beamsearch(score, beam_cut=0.0, beam_width=5, guided=True):
    """decodeutil.pyx:9-51.  `score`: torch tensor on the GPU (or a numpy array, uploaded) of
    shape (T, ntrans) or (T, N, ntrans).  Returns (int8 flip-flop state sequence, score) resp.
    ([sequences], scores (N,) float32)."""
    if not torch.is_tensor(score):
        if not torch.cuda.is_available():
            raise RuntimeError("beamsearch: no AMD GPU; the flip-flop operators only run as HIP kernels "
                               "(no CPU fallback)")
        score = torch.as_tensor(np.ascontiguousarray(score, dtype=np.float32)).cuda()
    _lib.require_gpu(score, "beamsearch")
    single = score.dim() == 2
    sc = (score.unsqueeze(1) if single else score).detach().float().contiguous()
    T, N, S = sc.shape
    nbase = flipflopfings.nbase_flipflop(S)
    if not 1 <= int(beam_width) <= 12 or int(beam_width) * (nbase + 1) > 64:
        raise ValueError("beamsearch: beam_width %d not in 1..12 (one candidate record per lane of a wavefront: "
                         "beam_width * (nbase + 1) <= 64)" % int(beam_width))
    if not 0.0 <= float(beam_cut) <= 1.0:
        raise ValueError("beamsearch: beam_cut %r outside [0, 1]" % (beam_cut,))
    L = _lib.lib()
    dev = sc.device
    with torch.cuda.device(dev):
        seq = torch.empty((N, T), dtype=torch.int8, device=dev)
        seqlen = torch.empty(N, dtype=torch.int32, device=dev)
        out = torch.empty(N, dtype=torch.float32, device=dev)
        wsb = L.tk_flipflop_beamsearch_workspace_bytes(T, N, nbase)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = L.tk_flipflop_beamsearch_dev(_lib.ptr(sc), T, N, nbase, int(beam_width), float(beam_cut),
                                          int(bool(guided)), _lib.ptr(seq), _lib.ptr(seqlen), _lib.ptr(out),
                                          _lib.ptr(ws), wsb, _lib.stream_ptr())
        _lib.check(rc, "tk_flipflop_beamsearch_dev")
    seq_h, len_h, sc_h = seq.cpu().numpy(), seqlen.cpu().numpy(), out.cpu().numpy()
    seqs = [seq_h[n, :len_h[n]].copy() for n in range(N)]
    if single:
        return seqs[0], float(sc_h[0])
    return seqs, sc_h


def _lattice(score, init, forward_pass, what):
    if not torch.is_tensor(score):
        if not torch.cuda.is_available():
            raise RuntimeError("%s: no AMD GPU; the flip-flop operators only run as HIP kernels "
                               "(no CPU fallback)" % what)
        score = torch.as_tensor(np.ascontiguousarray(score, dtype=np.float32)).cuda()
    _lib.require_gpu(score, what)
    single = score.dim() == 2
    sc = (score.unsqueeze(1) if single else score).detach().float().contiguous()
    T, N, S = sc.shape
    nbase = flipflopfings.nbase_flipflop(S)
    dev = sc.device
    with torch.cuda.device(dev):
        init_d = None
        if init is not None:
            init_d = torch.as_tensor(np.asarray(init, dtype=np.float32) if not torch.is_tensor(init) else init,
                                     dtype=torch.float32).to(dev).reshape(-1, 2 * nbase)
            if init_d.shape[0] == 1 and N > 1:
                init_d = init_d.expand(N, -1)
            init_d = init_d.contiguous()
            assert init_d.shape == (N, 2 * nbase), "init: one vector of 2 nbase states (per read)"
        out = torch.empty((N, T + 1, 2 * nbase), dtype=torch.float32, device=dev)
        total = torch.empty(N, dtype=torch.float32, device=dev)
        rc = _lib.lib().tk_flipflop_lattice_dev(_lib.ptr(sc), T, N, nbase, int(forward_pass), _lib.ptr(init_d),
                                                _lib.ptr(out), _lib.ptr(total), _lib.stream_ptr())
        _lib.check(rc, "tk_flipflop_lattice_dev")
    out_h, tot_h = out.cpu().numpy(), total.cpu().numpy()
    if single:
        return out_h[0], float(tot_h[0])
    return out_h, tot_h


def forward(score, init=None):
    """decodeutil.pyx:82-108 / c_flipflopfwdbwd.c:112-152: forward scores (T + 1, 2 nbase) of every
    block (row 0 = `init` or zeros) and their final log-sum-exp.  A batch (T, N, ntrans) returns
    (N, T + 1, 2 nbase) and (N,)."""
    return _lattice(score, init, True, "decodeutil.forward")


def backward(score, init=None):
    """decodeutil.pyx:54-79 / c_flipflopfwdbwd.c:55-91: backward scores (row T = `init` or zeros)."""
    return _lattice(score, init, False, "decodeutil.backward")
