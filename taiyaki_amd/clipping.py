"""Gradient clipping of the flip-flop trainer without host synchronisation.

Counterpart of ``apply_clipping`` (bin/train_flipflop.py:201-212) and
``maths.RollingMAD`` (taiyaki/maths.py:138-195).  The reference takes
``float(torch.max(torch.abs(p.grad)))`` per parameter tensor -- about 45 reductions and as
many host syncs per step -- feeds the maxima to a rolling median + MAD and clamps with the
resulting thresholds from iteration `window` on.  Here one HIP launch pair over the flat
gradient arena (csrc/clip_kernels.hip) produces all maxima and applies the thresholds; the
maxima travel to the host asynchronously and feed the thresholds of the NEXT step, which is
exactly the reference's data flow (train_flipflop.py:575-578).
"""
import numpy as np
import torch

from taiyaki_amd import _lib

MAD_SD_FACTOR = 1.4826      # maths.py:5


def med_mad(data, factor=None, axis=None):
    """maths.py:8-32: median and scaled median absolute deviation."""
    factor = MAD_SD_FACTOR if factor is None else factor
    dmed = np.median(data, axis=axis, keepdims=True)
    dmad = factor * np.median(np.abs(data - dmed), axis=axis, keepdims=True)
    if axis is None:
        return dmed.flatten()[0], dmad.flatten()[0]
    return dmed.squeeze(axis), dmad.squeeze(axis)


class RollingMAD:
    """maths.py:138-195: per-parameter `median + n_mads * MAD` over the last `window` values;
    `default_to` until `window` values have been seen."""

    def __init__(self, nparams, n_mads=0, window=1000, default_to=None):
        self.n_mads = n_mads
        self.default_to = default_to
        self._window_data = np.empty((nparams, window), dtype="f4")
        self._curr_iter = 0

    @property
    def nparams(self):
        return self._window_data.shape[0]

    @property
    def window(self):
        return self._window_data.shape[1]

    def update(self, vals):
        if len(vals) != self.nparams:
            raise AssertionError("Number of values (%d) provided does not match number of parameters "
                                 "(%d)." % (len(vals), self.nparams))
        self._window_data[:, self._curr_iter % self.window] = vals
        self._curr_iter += 1
        if self._curr_iter < self.window:
            return self.default_to
        med, mad = med_mad(self._window_data, axis=1)
        return med + mad * self.n_mads


class DeviceClipper:
    """Maxima + clip-by-value over a `parallel.FlatGradArena`, thresholds from a RollingMAD.

    `step()` = collect the maxima of the previous call (their copy finished long ago),
    update the thresholds, enqueue this step's kernels and the async copy of the new maxima.
    Returns the previous step's maxima (numpy) or None on the first call."""

    def __init__(self, arena, n_mads=0, window=1000):
        self.arena = arena
        dev = arena.flat.device
        sizes = [p.numel() for p in arena.params]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.nseg, self.max_len = len(sizes), int(max(sizes))
        self.seg_off = torch.from_numpy(off).to(dev)
        self.maxs = torch.zeros(self.nseg, dtype=torch.float32, device=dev)
        # +inf = "no threshold yet": the clamp kernel skips such segments, so it can always be
        # launched (and captured into a hipGraph) and starts clamping the moment thresholds arrive
        self.thresh = torch.full((self.nseg,), float("inf"), dtype=torch.float32, device=dev)
        self.maxs_host = torch.zeros(self.nseg, dtype=torch.float32).pin_memory() if dev.type == "cuda" \
            else torch.zeros(self.nseg, dtype=torch.float32)
        self.thresh_host = torch.zeros(self.nseg, dtype=torch.float32).pin_memory() if dev.type == "cuda" \
            else torch.zeros(self.nseg, dtype=torch.float32)
        self.rolling = RollingMAD(self.nseg, n_mads, window)
        self.event = None
        self.h2d_done = None
        self.active = False         # thresholds exist (window filled)
        self.last_maxs = None

    def collect(self):
        """Fold the previous step's maxima into the rolling statistics."""
        if self.event is None:
            return None
        self.event.synchronize()
        self.event = None
        self.last_maxs = self.maxs_host.numpy().copy()
        th = self.rolling.update(self.last_maxs)
        if th is not None:
            if self.h2d_done is not None:
                self.h2d_done.synchronize()         # the pinned buffer is free again
            self.thresh_host.copy_(torch.from_numpy(np.asarray(th, dtype=np.float32)))
            self.thresh.copy_(self.thresh_host, non_blocking=True)
            if self.thresh.is_cuda:
                self.h2d_done = torch.cuda.Event()
                self.h2d_done.record()
            self.active = True
        return self.last_maxs

    def launch_kernels(self):
        """Only the device work (capturable into a hipGraph): maxima, then the clamp (a no-op for
        segments whose threshold is still +inf).  A whole-step graph replays this; the host side
        of the step is `collect()` before the replay and `copy_maxima_async()` after it."""
        _lib.require_gpu(self.arena.flat, "DeviceClipper")
        L = _lib.lib()
        with torch.cuda.device(self.arena.flat.device):
            rc = L.tk_grad_maxabs_clip_dev(_lib.ptr(self.arena.flat), _lib.ptr(self.seg_off), self.nseg,
                                           self.max_len, _lib.ptr(self.thresh), _lib.ptr(self.maxs),
                                           _lib.stream_ptr())
            _lib.check(rc, "tk_grad_maxabs_clip_dev")

    def copy_maxima_async(self):
        """The maxima of the step just enqueued (or replayed from a graph) start their way to the
        host; `collect()` picks them up before the next step."""
        with torch.cuda.device(self.arena.flat.device):
            self.maxs_host.copy_(self.maxs, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()

    def launch(self):
        """Enqueue maxima (+ clamp once thresholds exist) and the async copy of the maxima."""
        self.launch_kernels()
        self.copy_maxima_async()

    def step(self):
        prev = self.collect()
        self.launch()
        return prev
