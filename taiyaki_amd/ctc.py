"""Drop-in for ``taiyaki.ctc`` (taiyaki/ctc/ctc.pyx): the flip-flop CRF losses.

Same names, positional signatures, argument meaning and error behaviour as the
reference's ``torch.autograd.Function``s, but ``forward`` launches the gfx950 HIP
kernels through the C ABI on ``torch.cuda.current_stream()``: no device->host copy
of the score tensor, no Python index building, no host->device copy of the
gradient (ctc.pyx:119, 127-132, 139-141 in the reference).
"""
import ctypes
import os

import numpy as np
import torch

from taiyaki_amd import _lib, flipflopfings


def _indices(seqs, seqlen, nbase, device, mod_cats=None, can_mods_offsets=None,
             mod_cat_weights=None, status=None, defer=False):
    """Device-side move/stay(/mod) ids in the padded per-position layout.  `defer`: allocate the arrays and
    return the `tk_seq_labels` as the 8th element instead of launching the build -- the operator's `_labels_dev`
    entry point builds them inside its first launch (round 5: one launch less per call)."""
    L = _lib.lib()
    seqs_d = seqs.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()
    # (an empty batch of labels: data_ptr() of an empty tensor is NULL, and the C side takes seqs == NULL with total_len == 0)
    seqlen_d = seqlen.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()
    nbatch = seqlen_d.numel()
    total = seqs_d.numel()
    seqoff = torch.empty(nbatch + 1, dtype=torch.int64, device=device)
    narr = 4 if mod_cats is not None else 2
    if defer:
        # the index arrays are SCRATCH of the labels entry points (written only by calls the linear path does not take):
        # views of one cached per-(device, stream) buffer instead of up to four allocations per call
        cap = max(total, 1)
        scratch = _workspace(narr * cap * 4, device, "idx")
        arrs = [scratch[k * cap * 4:(k + 1) * cap * 4] for k in range(narr)]
        stay, move = arrs[0].view(torch.int32), arrs[1].view(torch.int32)
    else:
        stay = torch.empty(max(total, 1), dtype=torch.int32, device=device)
        move = torch.empty(max(total, 1), dtype=torch.int32, device=device)
    mod = fact = mc = cmo = mcw = None
    if mod_cats is not None:
        mc = mod_cats.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()
        cmo = _device_constant(can_mods_offsets, torch.int32, device)
        mcw = _device_constant(mod_cat_weights, torch.float32, device)
        if defer:
            mod, fact = arrs[2].view(torch.int32), arrs[3].view(torch.float32)
        else:
            mod = torch.empty(max(total, 1), dtype=torch.int32, device=device)
            fact = torch.empty(max(total, 1), dtype=torch.float32, device=device)
    if defer:
        labels = _lib.SeqLabels(_lib.ptr(seqs_d), total, nbase, _lib.ptr(mc), _lib.ptr(cmo), _lib.ptr(mcw),
                                _bulk_seqlen(seqlen))
        return seqlen_d, seqoff, stay, move, mod, fact, (seqs_d, mc, cmo, mcw), labels
    rc = L.tk_flipflop_build_indices_dev(
        _lib.ptr(seqs_d), _lib.ptr(seqlen_d), nbatch, total, nbase, _lib.ptr(mc),
        _lib.ptr(cmo), _lib.ptr(mcw), _lib.ptr(seqoff), _lib.ptr(stay), _lib.ptr(move),
        _lib.ptr(mod), _lib.ptr(fact), _lib.ptr(status), _lib.stream_ptr())
    _lib.check(rc, "tk_flipflop_build_indices_dev")
    # keep the staging tensors alive until the caller has enqueued its kernel
    return seqlen_d, seqoff, stay, move, mod, fact, (seqs_d, mc, cmo, mcw)


_CONSTANTS = {}


def _device_constant(values, dtype, device):
    """The few-element cat-mod tables (can_mods_offsets, mod_cat_weights * mod_factor) on the
    device.  Tensors that already live there pass through; host values are uploaded once per
    distinct content (so a captured step never contains a pageable host-to-device copy)."""
    if torch.is_tensor(values) and values.device == device:
        return values.to(dtype).contiguous()
    arr = np.ascontiguousarray(values.cpu().numpy() if torch.is_tensor(values) else np.asarray(values))
    key = (str(device), str(dtype), arr.dtype.str, arr.tobytes())
    if key not in _CONSTANTS:
        if len(_CONSTANTS) > 256:
            _CONSTANTS.clear()
        _CONSTANTS[key] = torch.as_tensor(arr).to(dtype).to(device)
    return _CONSTANTS[key]


def n_mod_columns(can_mods_offsets):
    """can_mods_offsets[-1] as a host int.  The table must live on the host (numpy, list or CPU
    tensor): reading it off a device tensor would be a sync per call and is illegal while a
    hipGraph is being captured."""
    if torch.is_tensor(can_mods_offsets):
        if can_mods_offsets.is_cuda:
            raise TypeError("can_mods_offsets must be a host array (numpy / list / CPU tensor): its last entry "
                            "sizes the launch")
        return int(can_mods_offsets[-1])
    return int(np.asarray(can_mods_offsets)[-1])


def _col_weights(keep, mod):
    """The `mod_col_weights` argument of the C entry points: the device copy of `mod_cat_weights` that
    `tk_flipflop_build_indices_dev` filled `modfact` from -- the promise that a move's factor is a
    property of its modification column, which lets the kernels exponentiate a row once per wave.
    TK_CATMOD_GENERAL=1 withholds it (lab build only / tests: the general per-position form)."""
    if mod is None or _lab_switch("TK_CATMOD_GENERAL"):
        return None
    return _lib.ptr(keep[3])


_KEEP_WS = None      # debugging aid: set to a list to keep the kernels' workspaces alive


def _lab_switch(name):
    """A dispatch switch of this module (TK_CATMOD_GENERAL, TK_SEPARATE_INDEX_BUILD): read from the environment only
    while the process runs on the lab build -- like the kernels' own switches, the release path reads none."""
    return bool(_lib.is_lab() and os.environ.get(name))


def _workspace(nbytes, dev, tag):
    return _lib.workspace(nbytes, dev, tag, fresh=_KEEP_WS is not None)


release_workspaces = _lib.release_workspaces
last_gate_count = _lib.last_gate_count
last_retry_count = _lib.last_retry_count


def set_max_seqlen(seqlen, value, bulk=None):
    """Whoever assembles a batch knows its longest sequence on the host (bin/train_flipflop.py:133-138
    builds `seqlens` from Python lists); a `seqlens` tensor that lives on the device carries that
    number along as an attribute, so that the CRF launch is sized by it without a device sync
    (mapped_signal.sample_chunks, bench.make_batches, the graph trainers' static buffers).
    `bulk` (round 6, optional): a length all but a few of the batch's reads -- a sixteenth -- stay below
    (`bulk_of`); it picks the launch's block configuration (tk_seq_labels.bulk_seqlen).  Unknown: the fast
    configuration, and the few reads it may disown are retried one by one."""
    seqlen.tk_max_seqlen = int(value)
    seqlen.tk_bulk_seqlen = None if bulk is None else int(bulk)
    return seqlen


def bulk_of(lengths):
    """The batch's BULK length from its lengths on the host: the longest read once the longest sixteenth of the
    batch (at least one read) is set aside -- those few are what the retry launch has slots for."""
    a = np.sort(np.asarray(lengths).reshape(-1))
    if a.size == 0:
        return 0
    return int(a[max(0, a.size - 1 - max(1, a.size // 16))])


def _bulk_seqlen(seqlen):
    """tk_seq_labels.bulk_seqlen for this call: from the lengths where they live on the host, the hint of
    `set_max_seqlen` where they carry one, else 0 (unknown) -- never a device sync."""
    if not seqlen.numel():
        return 0
    if hasattr(seqlen, "tk_max_seqlen"):
        hint = getattr(seqlen, "tk_bulk_seqlen", None)
        return 0 if hint is None else int(hint)
    if seqlen.is_cuda:
        return 0
    return bulk_of(seqlen.numpy())


def _max_seqlen(seqlen):
    """Exact bound when seqlen lives on the host (bin/train_flipflop.py:133-138) or carries it
    (`set_max_seqlen`) -- no device sync.  For a bare device tensor (train_abinitio.py:207-210): in
    strict mode the call ends in a host sync anyway (the status word), so one more for the true
    maximum costs nothing and sizes the launch and its workspace by it; in non-strict mode 0
    (= unknown: sized for nblk + 1)."""
    if not seqlen.numel():
        return 0
    hint = getattr(seqlen, "tk_max_seqlen", None)
    if hint is not None:
        return int(hint)
    if seqlen.is_cuda and not _lib.is_strict():
        return 0
    return int(seqlen.max())


def _run(logprob, seqs, seqlen, sharp_can, sharp_mod, out_scale, ncan, want_grad,
         mod_cats=None, can_mods_offsets=None, mod_cat_weights=None):
    _lib.require_gpu(logprob, "flip-flop CRF loss")
    L = _lib.lib()
    lp = logprob.detach().float().contiguous()
    nblk, nbatch, ntrans = lp.shape
    nbase = flipflopfings.nbase_flipflop(ncan)
    dev = lp.device
    with torch.cuda.device(dev):
        status = _lib.status_word(dev)
        # TK_CATMOD_GENERAL=1 / TK_SEPARATE_INDEX_BUILD=1 (lab / tests): the stand-alone index kernel and the
        # entry point that takes its arrays; default: the index build rides in the operator's first launch
        separate = _lab_switch("TK_CATMOD_GENERAL") or _lab_switch("TK_SEPARATE_INDEX_BUILD")
        res = _indices(seqs, seqlen, nbase, dev, mod_cats, can_mods_offsets, mod_cat_weights, status, defer=not separate)
        seqlen_d, seqoff, stay, move, mod, fact, keep = res[:7]
        maxlen = _max_seqlen(seqlen)
        cost = torch.empty(nbatch, dtype=torch.float32, device=dev)
        grad = torch.empty_like(lp) if want_grad else None
        wsb = L.tk_crf_flipflop_workspace_bytes_sharp(ntrans, nblk, nbatch, maxlen, int(want_grad), float(sharp_can))
        ws = _workspace(wsb, dev, "crf")
        if _KEEP_WS is not None:
            ws.zero_()
        if separate:
            rc = L.tk_crf_flipflop_dev(
                _lib.ptr(lp), ntrans, nblk, nbatch, _lib.ptr(stay), _lib.ptr(move), _lib.ptr(mod),
                _lib.ptr(fact), _lib.ptr(seqlen_d), _lib.ptr(seqoff), maxlen, ncan,
                float(sharp_can), float(sharp_mod), float(out_scale), _lib.ptr(cost),
                _lib.ptr(grad), _lib.ptr(ws), wsb, _lib.ptr(status), _lib.stream_ptr(),
                _col_weights(keep, mod))     # (the per-column factors modfact was filled from)
            _lib.check(rc, "tk_crf_flipflop_dev")
        else:
            rc = L.tk_crf_flipflop_labels_dev(
                _lib.ptr(lp), ntrans, nblk, nbatch, ctypes.byref(res[7]), _lib.ptr(seqlen_d), _lib.ptr(seqoff),
                _lib.ptr(stay), _lib.ptr(move), _lib.ptr(mod), _lib.ptr(fact), maxlen, ncan,
                float(sharp_can), float(sharp_mod), float(out_scale), _lib.ptr(cost),
                _lib.ptr(grad), _lib.ptr(ws), wsb, _lib.ptr(status), _lib.stream_ptr())
            _lib.check(rc, "tk_crf_flipflop_labels_dev")
        _lib.finish(status)
    del keep
    if _KEEP_WS is not None:
        _KEEP_WS.append(ws)
    return cost, grad


# ---------------------------------------------------------------------------------------------
# the numpy-level functions of taiyaki.ctc (ctc.pyx:13-113, 162-255) on the exact C prototypes
# ---------------------------------------------------------------------------------------------
def nstate_to_nbase(nstate):
    """ctc.pyx:13-21: number of bases of a flip-flop model with `nstate` transition scores;
    AssertionError when nstate is not 2 nb (nb + 1)."""
    nbase_f = np.float32(np.sqrt(0.25 + (0.5 * nstate)) - 0.5)
    assert np.mod(nbase_f, 1) == 0, (
        'Number of states not valid for flip-flop model. ' +
        'nstates: {}\tconverted nbases: {}').format(nstate, nbase_f)
    return int(nbase_f)


def _host_call(name, logprob, idx_arrays, seqlen, want_grad, pin, check_states):
    """One call of an exact-prototype entry point (`crf_flipflop_cost` ... in
    include/taiyaki_amd_flipflop.h: host pointers in the reference's layout, staged through device
    memory by the library) with the Cython layer's contract around it: C-contiguous typed arrays,
    finite input, finite output, outputs allocated here (pinned on request), -x / nblk returned.
    `idx_arrays`: (array, dtype, argument name) in prototype order."""
    logprob = _typed(logprob, np.float32, 3, "logprob")
    seqlen = _typed(seqlen, np.int32, 1, "seqlen")
    idx = [_typed(a, dt, 1, nm) for a, dt, nm in idx_arrays]
    assert np.all(np.isfinite(logprob)), "Input not finite"
    nblk, nbatch, nstate = logprob.shape
    if check_states:
        nstate_to_nbase(nstate)
    if seqlen.shape[0] != nbatch:
        raise ValueError("seqlen has %d entries for a batch of %d" % (seqlen.shape[0], nbatch))
    # the C layer trusts the index arrays' lengths (c_crf_flipflop.c:446-451); a short array would be
    # read past its end by the staging loop, so check here what Cython's bounds checks cannot
    nstay = int(seqlen.sum())
    nmove = int(np.maximum(seqlen.astype(np.int64) - 1, 0).sum())
    for a, (_, _, nm) in zip(idx, idx_arrays):
        need = nstay if nm == "stayidxs" else nmove
        if a.shape[0] < need:
            raise ValueError("%s has %d entries, the sequence lengths need %d" % (nm, a.shape[0], need))
    costs = torch.zeros(nbatch, device='cpu', dtype=torch.float)
    grads = torch.zeros(nblk, nbatch, nstate, device='cpu', dtype=torch.float) if want_grad else None
    if pin:
        costs = costs.pin_memory()
        grads = grads.pin_memory() if want_grad else None
    args = [_lib._vp(logprob.ctypes.data), nstate, nblk, nbatch] + [_lib._vp(a.ctypes.data) for a in idx]
    args += [_lib._vp(seqlen.ctypes.data), _lib._vp(costs.data_ptr())]
    if want_grad:
        args.append(_lib._vp(grads.data_ptr()))
    getattr(_lib.lib(), name)(*args)
    costs_np = costs.numpy()
    assert np.all(np.isfinite(costs_np)), (
        "Error: all costs must be finite, got {}.\n"
        "Try restarting from a checkpoint with a lower learning rate."
    ).format(costs_np)
    if not want_grad:
        return -costs / nblk
    assert np.all(np.isfinite(grads.numpy())), ("Error: Gradients not finite.\n"
                                                "Try restarting from a checkpoint with a lower learning rate.")
    return -costs / nblk, -grads / nblk


def _typed(a, dtype, ndim, name):
    """The typed-buffer check of a Cython signature (`np.ndarray[np.float32_t, ndim=3, mode="c"]`):
    wrong dtype / rank / layout is an error there, not a silent conversion."""
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)" % (name, type(a).__name__))
    if a.dtype != np.dtype(dtype):
        raise ValueError("Buffer dtype mismatch for '%s', expected '%s' but got '%s'" % (name, np.dtype(dtype), a.dtype))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions for '%s' (expected %d, got %d)" % (name, ndim, a.ndim))
    if not a.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous ('%s')" % name)
    return a


def crf_flipflop_cost(logprob, moveidxs, stayidxs, seqlen, pin=False):
    """ctc.pyx:31-66.  logprob (nblk, nbatch, nstate) float32, moveidxs (sum(seqlen) - nseq) and
    stayidxs (sum(seqlen)) uintp transition ids, seqlen (nseq) int32, all C-contiguous numpy
    arrays; returns the costs -score / nblk as a CPU tensor (nbatch,)."""
    return _host_call("crf_flipflop_cost", logprob, [(moveidxs, np.uintp, "moveidxs"), (stayidxs, np.uintp, "stayidxs")],
                      seqlen, False, pin, True)


def crf_flipflop_grad(logprob, moveidxs, stayidxs, seqlen, pin=False):
    """ctc.pyx:69-113: (costs (nbatch,), d cost / d logprob (nblk, nbatch, nstate)) as CPU tensors."""
    return _host_call("crf_flipflop_grad", logprob, [(moveidxs, np.uintp, "moveidxs"), (stayidxs, np.uintp, "stayidxs")],
                      seqlen, True, pin, True)


def cat_mod_flipflop_cost(logprob, moveidxs, stayidxs, modmoveidxs, modmovefacts, seqlen, pin=False):
    """ctc.pyx:162-204: as `crf_flipflop_cost` with a modification column id (uintp) and factor
    (float32) per move."""
    return _host_call("cat_mod_flipflop_cost", logprob,
                      [(moveidxs, np.uintp, "moveidxs"), (stayidxs, np.uintp, "stayidxs"),
                       (modmoveidxs, np.uintp, "modmoveidxs"), (modmovefacts, np.float32, "modmovefacts")],
                      seqlen, False, pin, False)


def cat_mod_flipflop_grad(logprob, moveidxs, stayidxs, modmoveidxs, modmovefacts, seqlen, pin=False):
    """ctc.pyx:207-255"""
    return _host_call("cat_mod_flipflop_grad", logprob,
                      [(moveidxs, np.uintp, "moveidxs"), (stayidxs, np.uintp, "stayidxs"),
                       (modmoveidxs, np.uintp, "modmoveidxs"), (modmovefacts, np.float32, "modmovefacts")],
                      seqlen, True, pin, False)


class FlipFlopCRF(torch.autograd.Function):
    """taiyaki/ctc/ctc.pyx:116-151"""

    @staticmethod
    def forward(ctx, logprob, seqs, seqlen, sharpfact: float):
        ntrans = logprob.shape[2]
        # lp = sharp * logprob; returned cost/sharp; the saved gradient is
        # d(cost/sharp)/d logprob exactly (the two factors cancel, ctc.pyx:119,145)
        cost, grad = _run(logprob, seqs, seqlen, sharpfact, sharpfact, 1.0 / sharpfact,
                          ntrans, ctx.needs_input_grad[0])
        if grad is not None:
            ctx.save_for_backward(grad)
        return cost

    @staticmethod
    def backward(ctx, output_grads):
        grads, = ctx.saved_tensors
        return grads * output_grads.unsqueeze(1), None, None, None


crf_flipflop_loss = FlipFlopCRF.apply


class CatModFlipFlop(torch.autograd.Function):
    """taiyaki/ctc/ctc.pyx:258-310, including its quirk: only the canonical
    columns are sharpened (265-267) and backward returns the saved gradient
    d cost / d lp unscaled (306-310)."""

    @staticmethod
    def forward(ctx, logprob, seqs, seqlen, mod_cats, can_mods_offsets,
                mod_cat_weights, sharpfact: float):
        ntrans = logprob.shape[2]
        n_can_trans = ntrans - n_mod_columns(can_mods_offsets)
        cost, grad = _run(logprob, seqs, seqlen, sharpfact, 1.0, 1.0 / sharpfact,
                          n_can_trans, ctx.needs_input_grad[0], mod_cats,
                          can_mods_offsets, mod_cat_weights)
        if grad is not None:
            ctx.save_for_backward(grad)
        return cost

    @staticmethod
    def backward(ctx, output_grads):
        grads, = ctx.saved_tensors
        return (grads * output_grads.unsqueeze(1),
                None, None, None, None, None, None)


cat_mod_flipflop_loss = CatModFlipFlop.apply


# ---------------------------------------------------------------------------------------------
# fused train-step loss: (A) + (B) / nblk in one operator, one gradient tensor
# ---------------------------------------------------------------------------------------------
def _run_fused(outputs, seqs, seqlen, sharpfact, want_grad, grad_scale=1.0, grad_scale_per_read=None,
               mod_cats=None, can_mods_offsets=None, mod_cat_weights=None):
    _lib.require_gpu(outputs, "flip-flop loss")
    L = _lib.lib()
    lp = outputs.detach().float().contiguous()
    if lp.data_ptr() % 16 != 0:
        lp = lp.clone()
    nblk, nbatch, ntrans = lp.shape
    ncan = ntrans - (n_mod_columns(can_mods_offsets) if mod_cats is not None else 0)
    nbase = flipflopfings.nbase_flipflop(ncan)
    dev = lp.device
    with torch.cuda.device(dev):
        status = _lib.status_word(dev)
        separate = _lab_switch("TK_CATMOD_GENERAL") or _lab_switch("TK_SEPARATE_INDEX_BUILD")
        res = _indices(seqs, seqlen, nbase, dev, mod_cats, can_mods_offsets, mod_cat_weights, status=status,
                       defer=not separate)
        seqlen_d, seqoff, stay, move, mod, fact, keep = res[:7]
        maxlen = _max_seqlen(seqlen)
        lossvector = torch.empty(nbatch, dtype=torch.float32, device=dev)
        logz = torch.empty(nbatch, dtype=torch.float32, device=dev)
        grad = torch.empty_like(lp)
        wsa = L.tk_crf_flipflop_workspace_bytes_sharp(ntrans, nblk, nbatch, maxlen, 1, float(sharpfact))
        wsb = L.tk_flipflop_logz_workspace_bytes(nblk, nbatch, nbase)
        wsx = L.tk_flipflop_loss_fused_aux_bytes(nblk, nbatch, nbase, ntrans)
        if mod is None and torch.cuda.is_current_stream_capturing() and L.tk_flipflop_loss_overlap(-1) != 2:
            wsx = 0         # (a capture replays the one-queue form: no gradient buffer for kernel B in the graph's pool)
        ws_a = _workspace(wsa, dev, "crf")
        ws_b = _workspace(wsb, dev, "logz")
        ws_x = _workspace(wsx, dev, "aux") if wsx else None
        gvec = None
        if grad_scale_per_read is not None:
            gvec = grad_scale_per_read.detach().to(device=dev, dtype=torch.float32).contiguous()
        if separate:
            rc = L.tk_flipflop_loss_fused_dev(
                _lib.ptr(lp), nblk, nbatch, nbase, ntrans, _lib.ptr(stay), _lib.ptr(move), _lib.ptr(mod), _lib.ptr(fact),
                _lib.ptr(seqlen_d), _lib.ptr(seqoff), maxlen, float(sharpfact), float(grad_scale), _lib.ptr(gvec),
                _lib.ptr(lossvector), _lib.ptr(grad), _lib.ptr(logz),
                _lib.ptr(ws_a), wsa, _lib.ptr(ws_b), wsb, _lib.ptr(ws_x), wsx, _lib.ptr(status), _lib.stream_ptr(),
                _col_weights(keep, mod))
            _lib.check(rc, "tk_flipflop_loss_fused_dev")
        else:
            # (the index build rides in kernel A's first launch: one launch less in the captured step)
            rc = L.tk_flipflop_loss_fused_labels_dev(
                _lib.ptr(lp), nblk, nbatch, ntrans, ctypes.byref(res[7]), _lib.ptr(seqlen_d), _lib.ptr(seqoff),
                _lib.ptr(stay), _lib.ptr(move), _lib.ptr(mod), _lib.ptr(fact), maxlen, float(sharpfact),
                float(grad_scale), _lib.ptr(gvec), _lib.ptr(lossvector), _lib.ptr(grad), _lib.ptr(logz),
                _lib.ptr(ws_a), wsa, _lib.ptr(ws_b), wsb, _lib.ptr(ws_x), wsx, _lib.ptr(status), _lib.stream_ptr())
            _lib.check(rc, "tk_flipflop_loss_fused_labels_dev")
        _lib.finish(status)
    del keep, gvec, ws_x
    return lossvector, (grad if want_grad else None), logz


class FlipFlopLoss(torch.autograd.Function):
    """`crf_flipflop_loss(outputs, seqs, seqlens, sharpen) + flipflop_logpartition(outputs) / nblk`
    (the lossvector of bin/train_flipflop.py:172-182) as ONE operator: kernel B leaves
    d logZ / d outputs in the gradient tensor, kernel A's posterior pass adds its own term in
    place -- one (T, N, S) gradient tensor instead of two plus autograd's add."""

    @staticmethod
    def forward(ctx, outputs, seqs, seqlen, sharpfact: float, mod_cats=None, can_mods_offsets=None,
                mod_cat_weights=None):
        lossvector, grad, _ = _run_fused(outputs, seqs, seqlen, sharpfact, ctx.needs_input_grad[0],
                                         mod_cats=mod_cats, can_mods_offsets=can_mods_offsets,
                                         mod_cat_weights=mod_cat_weights)
        if grad is not None:
            ctx.save_for_backward(grad)
        return lossvector

    @staticmethod
    def backward(ctx, output_grads):
        grads, = ctx.saved_tensors
        return grads * output_grads.unsqueeze(1), None, None, None, None, None, None


flipflop_loss = FlipFlopLoss.apply


class FlipFlopMeanLoss(torch.autograd.Function):
    """`calculate_loss`'s last two lines folded into the operator (bin/train_flipflop.py:172-182):
    returns (loss, lossvector) with loss = sum_n w[n] lossvector[n]; `weights` None = the
    reference's `lossvector.mean()`, a (nbatch,) device tensor = any other fixed reduction (the mean
    over the non-empty columns of a padded batch).  The kernels write d loss / d outputs directly
    (their per-read gradient multiplier), so `backward` hands the saved tensor on: no elementwise
    pass over the (T, N, S) tensor between the loss and the network's backward.  `loss.backward()`
    sends grad_output = 1; any other value is honoured by one scaling pass."""

    @staticmethod
    def forward(ctx, outputs, seqs, seqlen, sharpfact: float, weights=None, mod_cats=None,
                can_mods_offsets=None, mod_cat_weights=None):
        """The last three arguments make it the cat-mod loss (ctc.pyx:258-312 on all columns + logZ
        of the canonical ones, bin/train_flipflop.py:165-176)."""
        nbatch = outputs.shape[1]
        mods = dict(mod_cats=mod_cats, can_mods_offsets=can_mods_offsets, mod_cat_weights=mod_cat_weights)
        if weights is None:
            lossvector, grad, _ = _run_fused(outputs, seqs, seqlen, sharpfact, ctx.needs_input_grad[0],
                                             grad_scale=1.0 / nbatch, **mods)
            loss = lossvector.mean()
        else:
            lossvector, grad, _ = _run_fused(outputs, seqs, seqlen, sharpfact, ctx.needs_input_grad[0],
                                             grad_scale_per_read=weights, **mods)
            loss = (lossvector * weights.to(lossvector.dtype)).sum()
        if grad is not None:
            ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(lossvector)
        return loss, lossvector

    @staticmethod
    def backward(ctx, grad_loss, _grad_lossvector):
        grads, = ctx.saved_tensors
        if getattr(ctx, "tk_unit_grad", False):     # (set by `backward_unit` on THIS node only)
            return grads, None, None, None, None, None, None, None
        return grads * grad_loss, None, None, None, None, None, None, None


def backward_unit(loss, **kwargs):
    """`loss.backward(**kwargs)` for a trainer that differentiates the mean loss as it came out of
    `flipflop_mean_loss`.  autograd then sends grad_output = 1 and the scaling pass over the
    (T, N, S) tensor is pointless: `backward` hands the kernels' tensor on untouched.  The shortcut
    is tied to the node, not to a mode: it is taken only when `loss` IS the operator's own output
    (its grad_fn is this operator's node).  A loss that went through anything else first --
    `loss * 2`, `loss / n_sub_batches`, a GradScaler, a sum of losses -- has another grad_fn, the
    flag is not set and the incoming gradient is honoured.  Nothing global, nothing another thread
    differentiating another loss can see."""
    node = loss.grad_fn
    ours = node is not None and type(node).__name__ == FlipFlopMeanLoss.__name__ + "Backward"
    if ours:
        node.tk_unit_grad = True
    try:
        loss.backward(**kwargs)
    finally:
        if ours:
            node.tk_unit_grad = False


flipflop_mean_loss = FlipFlopMeanLoss.apply
