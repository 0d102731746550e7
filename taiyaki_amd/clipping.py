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


def _middle(values, axis):
    """Median along `axis` by selection (np.partition at the one or two middle ranks); the mean of
    the two middle values is taken in the array's own floating type, which is what np.median gives."""
    x = np.asarray(values)
    if not np.issubdtype(x.dtype, np.floating):
        x = x.astype(np.float64)
    n = x.shape[axis]
    if n == 0:                                  # nothing to take the middle of: NaN, like np.median
        return np.full(np.delete(x.shape, axis), np.nan, dtype=x.dtype)[()]
    lo, hi = (n - 1) // 2, n // 2
    part = np.partition(x, (lo, hi) if hi != lo else lo, axis=axis)
    low = np.take(part, lo, axis=axis)
    return low if hi == lo else (low + np.take(part, hi, axis=axis)) * x.dtype.type(0.5)


def med_mad(data, factor=None, axis=None):
    """Robust location / scale of `data` (the interface of taiyaki/maths.py:8-32, which
    chunk_selection.py:98-131 and the clipping below call): the median, and the median distance
    from it times `factor` (default 1.4826: the MAD of a normal sample then estimates its standard
    deviation).  Over everything (`axis=None`: two scalars) or along one axis."""
    scale = MAD_SD_FACTOR if factor is None else factor
    x = np.asarray(data)
    if axis is None:
        x, axis = x.reshape(-1), 0
    centre = _middle(x, axis)
    spread = _middle(np.abs(x - np.expand_dims(centre, axis)), axis)
    return centre, scale * spread


class RollingMAD:
    """Clipping thresholds `median + n_mads * MAD` per parameter tensor over the last `window`
    gradient maxima (the interface of taiyaki/maths.py:138-195; `update` returns `default_to` until
    the window has filled).

    The window of every parameter is kept twice: as a ring in arrival order (which value expires
    next) and SORTED, updated per step by taking the expiring value out and putting the new one
    in -- two rank counts and a shift over (nparams, window) instead of re-sorting.  The median is
    then read off the middle of the sorted rows; only the deviations from it need one selection
    pass.  Values are float32 like the reference's window, so thresholds agree with it bit for bit
    (tests/golden/basecall_small.npz `rollingmad/*`).  A NaN maximum (a diverged step) is kept in the
    ring, sorts last, and makes that parameter's threshold NaN for as long as it is in the window;
    +inf takes part as an ordinary largest value (median and MAD stay finite unless half the window
    is infinite) -- both are what the reference's np.median over the raw window gives
    (taiyaki/maths.py:182-195)."""

    def __init__(self, nparams, n_mads=0, window=1000, default_to=None):
        self.n_mads = n_mads
        self.default_to = default_to
        self._ring = np.zeros((nparams, window), dtype=np.float32)
        self._sorted = np.full((nparams, window), np.inf, dtype=np.float32)   # unfilled slots sort last
        self._seen = 0

    @property
    def nparams(self):
        return self._ring.shape[0]

    @property
    def window(self):
        return self._ring.shape[1]

    def _swap_in(self, leaving, entering):
        """One value per row leaves the sorted rows, one enters; rows stay ascending."""
        rows = self._sorted
        width = rows.shape[1]
        col = np.arange(width)[None, :]
        at = (rows < leaving[:, None]).sum(axis=1)[:, None]             # rank of the value that leaves
        closed = np.where(col >= at, np.roll(rows, -1, axis=1), rows)    # gap closed; last column is free
        to = (closed[:, :width - 1] < entering[:, None]).sum(axis=1)[:, None]
        self._sorted = np.where(col < to, closed, np.where(col == to, entering[:, None], np.roll(closed, 1, axis=1)))

    def update(self, vals):
        vals = np.asarray(vals, dtype=np.float32).reshape(-1)
        if vals.shape[0] != self.nparams:
            raise AssertionError("RollingMAD.update: got %d values for %d parameters" % (vals.shape[0], self.nparams))
        key = np.where(np.isnan(vals), np.float32(np.inf), vals)        # sort key: NaN last, beside +inf
        slot = self._seen % self.window
        # before the window has filled the slot holds no value yet: an +inf placeholder leaves
        leaving = np.where(np.isnan(self._ring[:, slot]), np.float32(np.inf), self._ring[:, slot]) \
            if self._seen >= self.window else np.full(self.nparams, np.inf, dtype=np.float32)
        self._swap_in(leaving, key)
        self._ring[:, slot] = vals
        self._seen += 1
        if self._seen < self.window:
            return self.default_to
        width = self.window
        lo, hi = (width - 1) // 2, width // 2
        centre = self._sorted[:, lo] if lo == hi else (self._sorted[:, lo] + self._sorted[:, hi]) * np.float32(0.5)
        with np.errstate(invalid="ignore"):
            spread = _middle(np.abs(self._sorted - centre[:, None]), 1) * np.float32(MAD_SD_FACTOR)
            out = centre + spread * self.n_mads
        dirty = np.isnan(self._ring).any(axis=1)
        if dirty.any():
            out = np.where(dirty, np.float32(np.nan), out)
        return out


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
