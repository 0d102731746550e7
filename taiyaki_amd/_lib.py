"""ctypes binding of the C ABI in include/taiyaki_amd_flipflop.h.

PyTorch is plumbing here (device memory, streams); every hot-path computation
is a HIP kernel behind the C ABI.  There is NO CPU / eager fallback: if the
shared library is missing or a tensor is not on an AMD GPU the operators raise.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBNAME = "libtaiyaki_amd_flipflop.so"
LIBPATH = os.environ.get("TAIYAKI_AMD_LIB") or os.path.join(CSRC, LIBNAME)   # (override: lab builds)
# The LAB build of the same sources (-DTK_LAB): the only one that reads the dispatch's environment switches
# (TK_CRF_MODE, TK_CRF_BK, TK_LOGZ_CH, TK_CRF_NO_FALLBACK ...) and exports tk_lab_*.  tests/ and tools/ switch
# to it with `use_lab()` when they flip one; the operators never load it by themselves.
LAB_LIBNAME = "libtaiyaki_amd_flipflop_lab.so"
LAB_LIBPATH = os.environ.get("TAIYAKI_AMD_LAB_LIB") or os.path.join(CSRC, LAB_LIBNAME)    # (override: A/B builds under tools/lab/)

_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_f = ctypes.c_float
_i = ctypes.c_int

# symbol -> (restype, argtypes); mirrors include/taiyaki_amd_flipflop.h
SIGNATURES = {
    "tk_version": (ctypes.c_char_p, []),
    "tk_flipflop_build_indices_dev": (_i, [_vp, _vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp]),
    "tk_crf_flipflop_workspace_bytes": (_sz, [_sz, _sz, _sz, _sz, _i]),
    "tk_crf_flipflop_workspace_bytes_sharp": (_sz, [_sz, _sz, _sz, _sz, _i, _f]),
    "tk_crf_flipflop_dev": (_i, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz,
                                 _f, _f, _f, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "tk_crf_flipflop_labels_dev": (_i, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz,
                                        _f, _f, _f, _vp, _vp, _vp, _sz, _vp, _vp]),
    "tk_flipflop_loss_fused_labels_dev": (_i, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _vp, _vp,
                                              _vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp]),
    "tk_flipflop_loss_fused_aux_bytes": (_sz, [_sz, _sz, _sz, _sz]),
    "tk_flipflop_loss_overlap": (ctypes.c_int, [ctypes.c_int]),
    "tk_flipflop_loss_fused_dev": (_i, [_vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _vp, _vp,
                                       _vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "tk_flipflop_lattice_dev": (_i, [_vp, _sz, _sz, _sz, _i, _vp, _vp, _vp, _vp]),
    "tk_flipflop_beamsearch_workspace_bytes": (_sz, [_sz, _sz, _sz]),
    "tk_flipflop_beamsearch_dev": (_i, [_vp, _sz, _sz, _sz, _i, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "tk_flipflop_logz_workspace_bytes": (_sz, [_sz, _sz, _sz]),
    "tk_flipflop_logz_dev": (_i, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _sz, _vp, _vp]),
    "tk_flipflop_viterbi_workspace_bytes": (_sz, [_sz, _sz, _sz]),
    "tk_flipflop_viterbi_dev": (_i, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _sz, _vp]),
    "tk_flipflop_errprobs_dev": (_i, [_vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "tk_devcopy_f32_dev": (_i, [_vp, _vp, _sz, _vp]),
    "tk_grad_maxabs_clip_dev": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "tk_flipflop_remap_dev": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
    "tk_remap_path_to_ref_to_signal_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp]),
    "tk_chunks_locate_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tk_chunks_select_dev": (_i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    "tk_chunks_gather_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _i, _i, _sz, _vp, _vp,
                                  _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    # exact reference prototypes (host pointers)
    "crf_flipflop_grad": (None, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
    "crf_flipflop_cost": (None, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp]),
    "cat_mod_flipflop_grad": (None, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cat_mod_flipflop_cost": (None, [_vp, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
}

# libtaiyaki_amd_rccl.so (csrc/rccl_api.cpp): the gradient all-reduce straight on RCCL, for hosts
# that do not go through torch.distributed.  A library of its own; the Python trainers use
# ProcessGroupNCCL (the same RCCL calls) and never load it.
RCCL_LIBNAME = "libtaiyaki_amd_rccl.so"
RCCL_SIGNATURES = {
    "tk_rccl_unique_id_bytes": (_sz, []),
    "tk_rccl_unique_id": (_i, [_vp, _sz]),
    "tk_rccl_comm_init": (_i, [ctypes.POINTER(_vp), _i, _vp, _i]),
    "tk_rendezvous_bytes": (_i, [ctypes.c_char_p, _i, _i, _i, _vp, _sz, _i]),
    "tk_rccl_comm_init_rendezvous": (_i, [ctypes.POINTER(_vp), ctypes.c_char_p, _i, _i, _i, _i]),
    "tk_allreduce_f32_dev": (_i, [_vp, _vp, _sz, _vp]),
    "tk_broadcast_f32_dev": (_i, [_vp, _vp, _sz, _i, _vp]),
    "tk_rccl_comm_destroy": (_i, [_vp]),
}

class SeqLabels(ctypes.Structure):
    """include/taiyaki_amd_flipflop.h: tk_seq_labels (what tk_flipflop_build_indices_dev takes, for the entry points
    that build their indices inside their first launch)."""
    _fields_ = [("seqs", _vp), ("total_len", _sz), ("nbase", _sz), ("mod_cats", _vp), ("can_mods_offsets", _vp),
                ("mod_cat_weights", _vp), ("bulk_seqlen", _sz)]


ERRORS = {1: "bad argument (NULL / shape / 16-byte alignment)",
          2: "unsupported nbase / ntrans / sequence length for this build",
          3: "workspace too small", 4: "HIP launch failure"}

LAB_SIGNATURES = {"tk_lab_crf_band_phase": (None, [_i])}

_lib = None
_handles = {}


def build(force=False):
    """Compile the gfx950 shared libraries in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC, "-j%d" % min(8, os.cpu_count() or 4)], check=True,
                   stdout=subprocess.DEVNULL)
    return LIBPATH


def _load(path, signatures):
    handle = _handles.get(path)
    if handle is None:
        if not os.path.exists(path):
            raise RuntimeError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the flip-flop operators)" % path)
        handle = ctypes.CDLL(path)
        for name, (res, args) in signatures.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _handles[path] = handle
    return handle


def lib():
    global _lib
    if _lib is None:
        _lib = _load(LIBPATH, SIGNATURES)
    return _lib


def use_lab(flag=True):
    """Route the operators of THIS process through the lab build (flag) or back through the release
    library.  Returns the handle now in use.  Lab switches (environment variables read per launch,
    tk_lab_*) only exist there; the two libraries keep separate side queues and caches."""
    global _lib
    _lib = _load(LAB_LIBPATH, dict(SIGNATURES, **LAB_SIGNATURES)) if flag else _load(LIBPATH, SIGNATURES)
    return _lib


def is_lab():
    return _lib is not None and _lib is _handles.get(LAB_LIBPATH)


_rccl = None


def rccl_lib():
    """The RCCL C-ABI library (loads librccl: only for hosts that want the collective without
    torch.distributed)."""
    global _rccl
    if _rccl is None:
        handle = ctypes.CDLL(os.path.join(CSRC, RCCL_LIBNAME))
        for name, (res, args) in RCCL_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _rccl = handle
    return _rccl


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (code %d)" % (what, ERRORS.get(rc, "unknown"), rc))


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "%s: tensor is on %s; the flip-flop operators only run as HIP kernels on an AMD GPU "
            "(no CPU fallback)" % (what, t.device))


def ptr(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def stream_ptr():
    return _vp(torch.cuda.current_stream().cuda_stream)


# --------------------------------------------------------------------------
# scratch memory of the kernels
# --------------------------------------------------------------------------
_WS_CACHE = {}


def workspace(nbytes, dev, tag, fresh=False):
    """The kernels' scratch memory.  Eager calls reuse ONE buffer per (device, stream, operator
    slot `tag`), grown when a call needs more -- calls on a stream are ordered, so nothing that is
    still being read is handed out again -- instead of a `torch.empty` of up to gigabytes per call.
    While a hipGraph is being captured (or `fresh`) the buffer is allocated anew: it then belongs
    to the graph's private pool and lives as long as the graph."""
    if fresh or torch.cuda.is_current_stream_capturing():
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, tag)
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        if len(_WS_CACHE) > 64:
            _WS_CACHE.clear()
        _WS_CACHE.pop(key, None)            # (release the old one before asking for the larger one)
        buf = None
        buf = _WS_CACHE[key] = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    return buf


def release_workspaces():
    """Drop the cached scratch buffers (they are re-created on demand)."""
    _WS_CACHE.clear()


# --------------------------------------------------------------------------
# non-finite reporting (reference: AssertionError in ctc.pyx:48,62-65,107-112)
# --------------------------------------------------------------------------
_strict = os.environ.get("TAIYAKI_AMD_STRICT", "1") != "0"
_deferred = {}


def is_strict():
    return _strict


def set_strict(flag):
    """strict (default): every operator call checks its device status word at once
    (one host sync, exactly the reference's error timing).  Non-strict: status
    words accumulate per device and are checked by `raise_if_nonfinite()`."""
    global _strict
    _strict = bool(flag)


def status_word(device):
    if _strict:
        return torch.zeros(1, dtype=torch.int32, device=device)
    key = (device.type, device.index)
    if key not in _deferred:
        _deferred[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _deferred[key]


_gated = {"last": 0, "total": 0, "last_retried": 0, "total_retried": 0}
_COUNT_MASK, _GATED_SHIFT, _RETRIED_SHIFT = 0xfff, 8, 20       # include/taiyaki_amd_flipflop.h: TK_STATUS_*_SHIFT


def last_gate_count():
    """Reads that the CRF's linear-domain path handed to its log-domain kernel (status word bits
    8-19, include/taiyaki_amd_flipflop.h): in strict mode of the most recent operator call, in
    non-strict mode between the last two `raise_if_nonfinite()` / `take_gate_count()` checks.
    Such reads are RIGHT but cost ~1000x a read on the linear path (1 ms at T = 800): a batch that
    keeps producing them (scores far outside the network's 5 tanh range, violent cat-mod logits)
    is losing time."""
    return _gated["last"]


def last_retry_count():
    """Reads that the batch's launch of the CRF's linear path disowned and the retry launch swept again,
    alone and at its conservative configuration (status word bits 20-31; round 6) -- counted like
    `last_gate_count`, which says how many of them failed there too.  A retried read costs about what
    two reads cost the batch's launch."""
    return _gated["last_retried"]


def _u32(bits):
    """The status words are int32 tensors; the kernels add their counts into bits 8-31 (unsigned)."""
    return int(bits) & 0xffffffff


def _counts(bits):
    return (bits >> _GATED_SHIFT) & _COUNT_MASK, (bits >> _RETRIED_SHIFT) & _COUNT_MASK


def _note(redone, retried):
    _gated["last"], _gated["last_retried"] = redone, retried
    _gated["total"] += redone
    _gated["total_retried"] += retried


def _note_gated(bits):
    bits = _u32(bits)
    _note(*_counts(bits))
    return int(bits) & 0xff


def take_gate_count():
    """Non-strict mode: read and clear the COUNTS of the deferred status words (one sync per
    device), leaving the error flags to `raise_if_nonfinite()`.  Returns the number of reads
    redone in the log domain since the last check (`last_retry_count()`: the reads retried)."""
    n = m = 0
    for t in _deferred.values():
        bits = _u32(t.item())
        redone, retried = _counts(bits)
        n, m = n + redone, m + retried
        if bits >> 8:
            # take out exactly what was read (one device op, wrap-around arithmetic): counts that kernels on
            # other streams add between the read and this op are kept, not cleared with the rest
            take = (bits >> 8) << 8
            t.sub_(take - (1 << 32) if take >= (1 << 31) else take)
    _note(n, m)
    return n


def gated_total():
    """Reads redone in the log domain since the process started, as far as the status words have
    been read (every call in strict mode; at `raise_if_nonfinite()` / `take_gate_count()` otherwise)."""
    return _gated["total"]


def retried_total():
    return _gated["total_retried"]


def _raise(bits):
    bits = _u32(bits) & 0xff
    if bits & 8:
        # the reference: `assert np.all(stayidxs >= 0) and ...` style index checks in ctc.pyx
        raise AssertionError("Error: sequence labels out of range for the flip-flop model (flip-flop code "
                             "outside [0, 2 nbase), modification category outside its base's range, or "
                             "sum(seqlen) larger than the label array)")
    if bits & 16:
        raise RuntimeError("a sequence is longer than the max_seqlen the CRF kernel was launched for")
    if bits & 4:
        raise RuntimeError("sequence buffer overflow: a chunk batch needed more than max_bases_per_chunk bases "
                           "per chunk (its `seqs` would be truncated)")
    if bits & 1:
        raise AssertionError("Error: all costs must be finite.\n"
                             "Try restarting from a checkpoint with a lower learning rate.")
    if bits & 2:
        raise AssertionError("Error: Gradients not finite.\n"
                             "Try restarting from a checkpoint with a lower learning rate.")


def finish(status):
    if _strict:
        _raise(_note_gated(int(status.item())))


def raise_if_nonfinite():
    """Check (and clear) the deferred status words; one sync per device."""
    total, retried, first_bad = 0, 0, 0
    for t in _deferred.values():
        bits = _u32(t.item())
        t.zero_()
        redone, again = _counts(bits)
        total, retried = total + redone, retried + again
        first_bad = first_bad or (bits & 0xff)
    _note(total, retried)
    _raise(first_bad)
