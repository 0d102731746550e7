"""Drop-in for ``taiyaki.flipflop_remap`` (taiyaki/flipflop_remap.py): the best alignment of a
matrix of flip-flop transition scores to a known sequence, used by prepare_mapped_reads
(prepare_mapping_funcs.py:88) -- on the GPU (csrc/remap_kernels.hip through the C ABI), for one
read or a whole batch of reads per launch.  Same names, argument meaning and return values:
``(score, path)`` with ``path`` of length T + 1, -1 where the alignment sits in the start / end
state.  float64 arithmetic like the reference, so scores and paths are identical to it.
"""
import numpy as np
import torch

from taiyaki_amd import _lib, flipflopfings

DEFAULT_ALPHABET = 'ACGT'       # taiyaki/constants.py
LARGE_VAL = 1e30


def remap_indices(sequence, alphabet=DEFAULT_ALPHABET):
    """The transition-score columns the alignment of `sequence` (str, or an array of base numbers)
    walks through: `(step_index (M-1), stay_index (M))`, what flipflop_remap.py:131-140 hands to
    `map_to_crf_viterbi`.  They are the loss's own transition ids (csrc/crf_kernels.hip
    `build_indices_kernel`, flipflopfings.py:6-31): flip-flop code the bases (second, fourth, ...
    base of a homopolymer run = flop), then stay = the code's self transition and step = the
    transition from one code into the next."""
    nbase = len(alphabet)
    if isinstance(sequence, str):
        lookup = {ch: k for k, ch in enumerate(alphabet)}
        bases = np.fromiter((lookup.get(ch, -1) for ch in sequence), dtype=np.int64, count=len(sequence))
    else:
        bases = np.asarray(sequence, dtype=np.int64)
    codes = flipflopfings.flipflop_code(bases, nbase)
    return flipflopfings.move_indices(codes, nbase), flipflopfings.stay_indices(codes, nbase)


def map_to_crf_viterbi_batch(scores, step_indices, stay_indices, localpen=LARGE_VAL, device=None):
    """map_to_crf_viterbi for several reads in one launch (one workgroup per read).

    scores: list of (T_i, K) float32 arrays or tensors (host or device); localpen: one float or
    one per read.
    Returns (scores float64 ndarray (nread,), [path_i int64 ndarray (T_i + 1,)])."""
    nread = len(scores)
    assert nread == len(step_indices) == len(stay_indices)
    if nread == 0:
        return np.zeros(0), []
    stay = [np.asarray(s, dtype=np.int64) for s in stay_indices]
    step = [np.asarray(s, dtype=np.int64) for s in step_indices]
    for st, sp in zip(stay, step):
        assert len(sp) == len(st) - 1           # flipflop_remap.py:28 (an empty sequence fails here too)
    if device is None:
        device = next((s.device for s in scores if torch.is_tensor(s) and s.is_cuda), torch.device("cuda"))
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("flipflop_remap runs as a HIP kernel on an AMD GPU; device=%s (no CPU fallback)" % device)
    K = int(scores[0].shape[1])
    for sc, st, sp in zip(scores, stay, step):
        assert sc.ndim == 2 and sc.shape[1] == K
        if len(st) and (st.max() >= K or st.min() < 0 or (len(sp) and (sp.max() >= K or sp.min() < 0))):
            raise IndexError("transition index out of range for %d transitions" % K)
    T = np.array([int(s.shape[0]) for s in scores], dtype=np.int64)
    M = np.array([len(s) for s in stay], dtype=np.int64)
    row_off = np.concatenate([[0], np.cumsum(T)])
    seq_off = np.concatenate([[0], np.cumsum(M)])
    tb_words = T * ((M + 63) // 64)
    tb_off = np.concatenate([[0], np.cumsum(tb_words)])
    with torch.cuda.device(device):
        as_dev = [torch.as_tensor(s).to(device=device, dtype=torch.float32) for s in scores]
        sc_d = torch.cat(as_dev, dim=0).contiguous() if nread > 1 else as_dev[0].contiguous()
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)    # noqa: E731
        stay_d = up(np.concatenate(stay), torch.int32)
        step_cat = np.concatenate(step) if sum(len(s) for s in step) else np.zeros(1, dtype=np.int64)
        step_d = up(step_cat, torch.int32)
        pen = np.array(np.broadcast_to(np.asarray(localpen, dtype=np.float64), (nread,)))
        pen_d = up(pen, torch.float64)
        row_d, seq_d, tbo_d = up(row_off, torch.int64), up(seq_off, torch.int64), up(tb_off[:-1], torch.int64)
        score_d = torch.empty(nread, dtype=torch.float64, device=device)
        path_d = torch.empty(int(row_off[-1]) + nread, dtype=torch.int64, device=device)
        tb_d = torch.empty(max(int(tb_off[-1]), 1), dtype=torch.int64, device=device)
        rc = _lib.lib().tk_flipflop_remap_dev(
            _lib.ptr(sc_d), _lib.ptr(row_d), K, _lib.ptr(stay_d), _lib.ptr(step_d), _lib.ptr(seq_d),
            _lib.ptr(pen_d), nread, int(M.max()), _lib.ptr(score_d), _lib.ptr(path_d), _lib.ptr(tb_d),
            _lib.ptr(tbo_d), _lib.stream_ptr())
        _lib.check(rc, "tk_flipflop_remap_dev")
        score = score_d.cpu().numpy()
        path = path_d.cpu().numpy()
    return score, [path[row_off[i] + i:row_off[i + 1] + i + 1] for i in range(nread)]


def map_to_crf_viterbi(scores, step_index, stay_index, localpen=LARGE_VAL):
    """flipflop_remap.py:6-88.  Returns (score of best path, best path)."""
    score, paths = map_to_crf_viterbi_batch([scores], [step_index], [stay_index], localpen)
    return float(score[0]), paths[0]


def flipflop_remap(transition_scores, sequence, alphabet=DEFAULT_ALPHABET, localpen=LARGE_VAL):
    """flipflop_remap.py:91-143.  Returns (alignment score, sequence positions (T + 1))."""
    step_index, stay_index = remap_indices(sequence, alphabet)
    return map_to_crf_viterbi(transition_scores, step_index, stay_index, localpen=localpen)


def flipflop_remap_batch(transition_scores, sequences, alphabet=DEFAULT_ALPHABET, localpen=LARGE_VAL):
    """flipflop_remap for a list of reads in one launch."""
    idx = [remap_indices(s, alphabet) for s in sequences]
    return map_to_crf_viterbi_batch(transition_scores, [i[0] for i in idx], [i[1] for i in idx], localpen)


def ref_to_signal_from_remapping_paths(paths, reflens, stride, signalstarts, siglens, device=None):
    """``SignalMapping.from_remapping_path(...).Ref_to_signal`` (signal_mapping.py:268-316 with
    ``get_reftosignal`` :202-265) for a batch of reads on the device.

    paths: list of (T_i + 1,) remapping paths (-1 at the clipped ends, non-decreasing in between --
    what ``flipflop_remap`` returns); reflens: reference lengths; stride: model stride;
    signalstarts / siglens: ``sig.signalstart`` and ``len(sig.untrimmed_dacs)`` per read.
    Returns a list of int32 arrays of length reflen_i + 1."""
    nread = len(paths)
    if nread == 0:
        return []
    host = [np.asarray(p.cpu() if torch.is_tensor(p) else p, dtype=np.int64) for p in paths]
    for p in host:
        body = p[p >= 0]
        inner = np.flatnonzero(p >= 0)
        if len(body) and (np.any(np.diff(body) < 0) or inner[-1] - inner[0] + 1 != len(inner) or p.min() < -1):
            raise ValueError("remapping path must be -1 at its ends and non-decreasing in between")
    device = torch.device("cuda" if device is None else device)
    if device.type != "cuda":
        raise RuntimeError("runs as a HIP kernel on an AMD GPU; device=%s (no CPU fallback)" % device)
    path_off = np.concatenate([[0], np.cumsum([len(p) for p in host])]).astype(np.int64)
    ref_off = np.concatenate([[0], np.cumsum(np.asarray(reflens, dtype=np.int64))]).astype(np.int64)
    with torch.cuda.device(device):
        up = lambda a: torch.from_numpy(np.array(a, dtype=np.int64)).to(device)    # noqa: E731
        path_d, po_d, ro_d = up(np.concatenate(host)), up(path_off), up(ref_off)
        ss_d = up(np.broadcast_to(np.asarray(signalstarts), (nread,)))
        sl_d = up(np.broadcast_to(np.asarray(siglens), (nread,)))
        out = torch.empty(int(ref_off[-1]) + nread, dtype=torch.int32, device=device)
        rc = _lib.lib().tk_remap_path_to_ref_to_signal_dev(
            _lib.ptr(path_d), _lib.ptr(po_d), _lib.ptr(ro_d), _lib.ptr(ss_d), _lib.ptr(sl_d), int(stride),
            nread, _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, "tk_remap_path_to_ref_to_signal_dev")
        flat = out.cpu().numpy()
    return [flat[ref_off[i] + i:ref_off[i + 1] + i + 1] for i in range(nread)]
