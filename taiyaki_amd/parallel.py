"""Single-node data parallelism: one process per GPU, reads sharded over ranks,
ONE flat-buffer gradient all-reduce per optimiser step over RCCL/xGMI.

Replaces the reference's DistributedDataParallel glue
(bin/train_flipflop.py:255-268 init, 384-397 wrap, all-reduce inside backward).
The payload is small (2,715,280 fp32 = 10.9 MB for mLstm size 256), so instead of
25 MiB bucketing every trainable gradient lives in one contiguous arena and is
reduced by a single `all_reduce(SUM)`; the 1/world scale is applied in the same
pass that the caller uses for clipping.  Works with backend "nccl" (= RCCL on
ROCm) and "gloo" (CPU tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, process_group=True):
    """Rendezvous from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    Returns (rank, local_rank, world).  `process_group=False`: read the environment only -- the
    caller brings the ranks together WITHOUT torch.distributed (`DirectRccl`'s own socket rendezvous,
    `RankComm`)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not process_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        return rank, local, world
    # TK_FORCE_PROCESS_GROUP=1: build the (single-rank) RCCL communicator anyway -- lets a 1-GPU
    # box exercise the collective path, its streams and the NCCL watchdog thread
    forced = bool(os.environ.get("TK_FORCE_PROCESS_GROUP"))
    if (world > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if os.environ.get("TK_ARENA_TRACE"):
            # RCCL then prints its topology and, per collective, the algorithm / protocol / channel
            # count it picked (ring vs tree, LL vs Simple) to stderr: the first thing to look at
            # when the 8-GPU curve disappoints
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL")
        if backend == "nccl":
            torch.cuda.set_device(local)
            # RCCL's kernels go to a HIGH-PRIORITY stream: HIP keeps a hardware queue per
            # priority level, so the collectives never share (or wait in) the queue of this
            # rank's compute streams, however few queues GPU_MAX_HW_QUEUES leaves those
            # (tools/queue_probe.py)
            opts = None
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
            except Exception:
                os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local), pg_options=opts)
        else:
            dist.init_process_group(backend=backend)
    return rank, local, world


def pin_rank(local, nlocal):
    """One core set per rank: the host side of a step is ~8,000 eager launches of the RNN
    backward from ONE thread per rank, and eight such threads (plus their OpenMP / autograd
    helper threads) wandering over the same cores cost each other cache and time.  Rank `local`
    of `nlocal` gets the `local`-th slice of the cores this process may use; intra-op threads are
    capped to that slice.  Returns the cores (sorted) -- reported in the bench line."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:          # not Linux
        return []
    per = max(1, len(cores) // max(1, nlocal))
    mine = cores[(local * per) % len(cores):][:per] or cores
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return cores
    torch.set_num_threads(max(1, min(len(mine), int(os.environ.get("OMP_NUM_THREADS", "8")))))
    return mine


def _trace(msg):
    """TK_ARENA_TRACE=1: every collective this rank issues, to stderr (debugging rank mismatches)."""
    if os.environ.get("TK_ARENA_TRACE"):
        import sys
        print("[arena rank %s] %s" % (os.environ.get("RANK", "?"), msg), file=sys.stderr, flush=True)


def rendezvous_port():
    """Where rank 0 serves RCCL's unique id: TK_RENDEZVOUS_PORT, else MASTER_PORT + 1 (MASTER_PORT itself
    belongs to torch.distributed.run's own store when the ranks were started by it)."""
    if os.environ.get("TK_RENDEZVOUS_PORT"):
        return int(os.environ["TK_RENDEZVOUS_PORT"])
    return int(os.environ.get("MASTER_PORT", "29500")) + 1


def socket_rendezvous(rank, world, payload, nbytes, addr=None, port=None, timeout_s=120.0):
    """Rank 0's `payload` (bytes of length `nbytes`) to every rank through the C ABI's
    `tk_rendezvous_bytes` (csrc/rccl_api.cpp: plain sockets at addr:port; no torch.distributed, no
    GPU).  Returns the bytes on every rank."""
    import ctypes
    from . import _lib
    L = _lib.rccl_lib()
    buf = ctypes.create_string_buffer(nbytes)
    if rank == 0:
        if payload is None or len(payload) != nbytes:
            raise ValueError("socket_rendezvous: rank 0 needs %d bytes to hand out" % nbytes)
        buf.raw = bytes(payload)
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = rendezvous_port() if port is None else int(port)
    rc = L.tk_rendezvous_bytes(addr.encode(), port, rank, world, buf, nbytes, int(timeout_s * 1000))
    if rc != 0:
        raise RuntimeError("rendezvous at %s:%d failed for rank %d of %d (code %d: %s)" % (
            addr, port, rank, world, rc, "bad argument" if rc == 1 else
            "timed out, a rank announced another world size, or a rank came twice"))
    return buf.raw


class _NoWork:
    def wait(self):
        pass


class _StreamWork:
    """What `DirectRccl.all_reduce` returns: `wait()` makes the CURRENT stream wait for the
    collective (the contract of a torch.distributed async work object on a GPU backend)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class DirectRccl:
    """The arena's collectives straight on RCCL through this repo's C ABI
    (csrc/rccl_api.cpp -> libtaiyaki_amd_rccl.so: `tk_rccl_comm_init`, `tk_allreduce_f32_dev`,
    `tk_broadcast_f32_dev`) instead of ProcessGroupNCCL -- the path a host without
    torch.distributed binds, and `TK_RCCL_DIRECT=1` for the Python trainers.  One communicator per
    process; the collectives are enqueued on a high-priority side stream of this class (a hardware
    queue of its own, like ProcessGroupNCCL's, tools/queue_probe.py) behind an event of the stream
    that produced the gradients.

    `exchange(id_bytes_or_None) -> id_bytes` hands rank 0's RCCL unique id to every rank (the
    reference's TCP store, bin/train_flipflop.py:255-268).  Default: when a torch.distributed group
    is up it carries the id; otherwise -- the path WITHOUT torch.distributed -- the C ABI's own socket
    rendezvous (`tk_rendezvous_bytes`: rank 0 serves the id at MASTER_ADDR : rendezvous_port(), plain
    sockets).  The identity for a single rank."""

    def __init__(self, rank, world, exchange=None, device=None, in_stream=False):
        """`in_stream`: enqueue every collective on the CALLER's current stream -- in order with the
        kernels that produced the gradients and with the optimiser behind them: no side stream, no
        events, no cross-queue signal.  Nothing is overlapped then; for this path's 10.9 MB payload
        there is nothing worth overlapping."""
        import ctypes
        from . import _lib
        self._lib = _lib.rccl_lib()
        self.rank, self.world = rank, world
        if device is not None:
            torch.cuda.set_device(device)
        nb = int(self._lib.tk_rccl_unique_id_bytes())
        buf = ctypes.create_string_buffer(nb)
        if rank == 0:
            _lib.check(self._lib.tk_rccl_unique_id(buf, nb), "tk_rccl_unique_id")
        if exchange is None:
            exchange = self._exchange_over_process_group if dist.is_initialized() else self._exchange_over_sockets
        idbytes = exchange(buf.raw if rank == 0 else None)
        if idbytes is None or len(idbytes) != nb:
            raise RuntimeError("DirectRccl: the unique id did not arrive (%r)" % (idbytes,))
        comm = ctypes.c_void_p()
        _lib.check(self._lib.tk_rccl_comm_init(ctypes.byref(comm), world, idbytes, rank), "tk_rccl_comm_init")
        self._comm = comm
        self.in_stream = bool(in_stream)
        self.stream = None if self.in_stream else torch.cuda.Stream(priority=-1)

    def _exchange_over_process_group(self, idbytes):
        if self.world == 1:
            return idbytes
        box = [idbytes]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def _exchange_over_sockets(self, idbytes):
        if self.world == 1:
            return idbytes
        return socket_rendezvous(self.rank, self.world, idbytes, nbytes=int(self._lib.tk_rccl_unique_id_bytes()))

    def _enqueue(self, fn, what, t):
        from . import _lib
        if self.in_stream:
            _lib.check(fn(_lib.stream_ptr()), what)
            return _NoWork()
        t.record_stream(self.stream)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)
        _lib.check(fn(_lib._vp(self.stream.cuda_stream)), what)
        done = torch.cuda.Event()
        done.record(self.stream)
        return _StreamWork(done)

    def all_reduce(self, t):
        """SUM over ranks, in place, asynchronously; `t`: contiguous float32 on this device."""
        from . import _lib
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return self._enqueue(lambda st: self._lib.tk_allreduce_f32_dev(self._comm, _lib.ptr(t), t.numel(), st),
                             "tk_allreduce_f32_dev", t)

    def broadcast(self, t, src=0):
        from . import _lib
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return self._enqueue(lambda st: self._lib.tk_broadcast_f32_dev(self._comm, _lib.ptr(t), t.numel(), src, st),
                             "tk_broadcast_f32_dev", t)

    def close(self):
        if self._comm is not None:
            torch.cuda.synchronize()
            self._lib.tk_rccl_comm_destroy(self._comm)
            self._comm = None


class RankComm:
    """What a data-parallel driver needs besides the gradient collective -- a barrier and one number per
    rank (max-over-ranks timing, "did every rank capture its graph") -- on either transport:
    `collective=None`: the torch.distributed group that is up (nccl or gloo);
    `collective=DirectRccl`: the C ABI's all-reduce alone (a slot per rank in a zero vector, SUM), so a job
    started with `bench.py --collective direct` never touches torch.distributed."""

    def __init__(self, rank, world, collective=None, device=None):
        self.rank, self.world, self.collective = rank, world, collective
        self.device = device if device is not None else torch.device("cpu")

    @property
    def active(self):
        return self.world > 1 and (self.collective is not None or dist.is_initialized())

    def gather(self, value):
        """`value` (a float) of every rank, in rank order, on every rank."""
        if not self.active:
            return [float(value)]
        if self.collective is not None:
            slots = torch.zeros(self.world, dtype=torch.float32, device=self.device)
            slots[self.rank] = float(value)
            self.collective.all_reduce(slots).wait()
            torch.cuda.current_stream().synchronize()
            return [float(v) for v in slots.tolist()]
        dev = self.device if dist.get_backend() == "nccl" else torch.device("cpu")
        mine = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        return [float(t.item()) for t in out]

    def barrier(self):
        if not self.active:
            return
        if self.collective is not None:
            self.gather(0.0)
        else:
            dist.barrier()


class FlatGradArena:
    """All trainable gradients as views into one contiguous fp32 buffer.

    Default (`overlap_buckets=0`): ONE all-reduce of the whole arena after backward.  That is the
    measured choice (profiles/r4_forced_group_bisect.txt, a one-rank RCCL group on one MI355X): the
    process group, its watchdog and one flat all-reduce cost the 109.4 ms step 0.1 ms; the same
    10.9 MB issued as 4-5 slices from backward hooks cost 16-19 ms, because every slice is a
    collective enqueued on a side queue in the middle of the host-bound eager RNN backward.  The
    payload needs ~0.1 ms on xGMI: there is nothing for an overlap to hide.

    `overlap_buckets=k` (k > 1, world > 1; kept for larger models) cuts the arena into k contiguous
    slices and all-reduces a slice as soon as autograd has accumulated the last gradient that lives
    in it (post-accumulate hooks): backward produces the gradients of the last layers first, so
    their reduction runs on RCCL's own stream while the RNN backward of the earlier layers is still
    computing (the reference's DDP does the same with 25 MiB buckets).  `finish()` waits for every
    slice, reduces whatever was not ready (unused parameters) and applies the 1/world factor."""

    def __init__(self, module, overlap_buckets=0, collective=None):
        """`collective`: a `DirectRccl` (this repo's C ABI over RCCL) instead of torch.distributed's
        process group; with it the arena reduces even for a single rank (a one-rank communicator
        is how a 1-GPU box exercises the path)."""
        self.collective = collective
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.spans = []
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self.spans.append((off, off + p.numel()))
            off += p.numel()
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._always = dist.is_initialized() and bool(os.environ.get("TK_FORCE_PROCESS_GROUP"))
        if collective is not None:
            self.world, self._always = collective.world, True
        self._work = []
        # TK_ARENA_LAB (tools/forced_group_bisect.py): "noop" = issue nothing (what the mere existence
        # of the process group costs), "hooks" = hooks installed and counted but no collective
        self._lab = os.environ.get("TK_ARENA_LAB", "")
        self._buckets = []          # [lo, hi, params still missing this step, issued]
        self._hooks = []
        self.hooks_enabled = True   # (switched off around a whole-step graph capture)
        if overlap_buckets > 1 and (self.world > 1 or self._always):
            self._install_hooks(overlap_buckets)

    # -- overlap ---------------------------------------------------------------------------
    def _install_hooks(self, nbuckets):
        per = -(-self.flat.numel() // nbuckets)
        bounds, lo, cnt = [], 0, []
        owner = []
        for (a, b) in self.spans:
            if b - lo > per and a > lo:
                bounds.append((lo, a))
                lo = a
            owner.append(len(bounds))
        bounds.append((lo, self.flat.numel()))
        cnt = [0] * len(bounds)
        for k in owner:
            cnt[k] += 1
        self._bucket_size = cnt
        self._buckets = [[a, b, c, False] for (a, b), c in zip(bounds, cnt)]
        for p, k in zip(self.params, owner):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(k)))

    def _make_hook(self, k):
        def hook(_param):
            if not self.hooks_enabled:
                return
            b = self._buckets[k]
            b[2] -= 1
            if b[2] == 0 and not b[3]:
                b[3] = True
                _trace("hook bucket %d [%d:%d]" % (k, b[0], b[1]))
                self._work.append(self._all_reduce(self.flat[b[0]:b[1]]))
        return hook

    def _all_reduce(self, t):
        if self._lab in ("noop", "hooks"):
            return _NoWork()
        if self.collective is not None:
            return self.collective.all_reduce(t)
        if self._lab == "sync":     # ProcessGroupNCCL on the CURRENT stream (torch >= 2.8: async_op=False)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return _NoWork()
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    @property
    def overlapped(self):
        return bool(self._buckets)

    def slices(self):
        """(lo, hi) of every all-reduce a step issues (one: the whole arena without overlap hooks)."""
        return [(b[0], b[1]) for b in self._buckets] or [(0, self.flat.numel())]

    # -- step interface ----------------------------------------------------------------------
    def zero(self):
        self.flat.zero_()

    def allreduce_async(self):
        """SUM over ranks (the reference's DDP averages; the 1/world factor is applied by
        `finish`).  With overlap hooks installed only the slices backward did not complete are
        issued here."""
        if not (self.world > 1 or self._always):
            return
        if self._buckets and self.hooks_enabled:
            for k, b in enumerate(self._buckets):
                if not b[3]:
                    b[3] = True
                    _trace("late bucket %d [%d:%d] missing %d" % (k, b[0], b[1], b[2]))
                    self._work.append(self._all_reduce(self.flat[b[0]:b[1]]))
        else:
            _trace("whole arena")
            self._work.append(self._all_reduce(self.flat))

    def finish(self, scale=1.0):
        """Wait for the slices, apply 1 / world (x `scale`: 1 / number of accumulated sub-batches)."""
        _trace("finish: %d pending" % len(self._work))
        if self._work:
            for w in self._work:
                w.wait()
            self._work = []
            scale = scale / self.world
        if scale != 1.0:
            self.flat.mul_(scale)
        for b, c in zip(self._buckets, getattr(self, "_bucket_size", [])):
            b[2], b[3] = c, False


def broadcast_parameters(module, src=0, collective=None):
    """Replaces the checkpoint-file + barrier handshake of the reference
    (bin/train_flipflop.py:380-392): rank 0's weights and buffers go out over RCCL.  With
    `collective` the contiguous float32 device tensors (every parameter and floating-point buffer
    of the models here) travel through the C ABI's `tk_broadcast_f32_dev`; anything else (integer
    buffers, other dtypes, non-contiguous storage) goes through torch.distributed when a group is
    up and is an ERROR otherwise -- never skipped: ranks must not start from different values."""
    tensors = list(module.parameters()) + list(module.buffers())
    if collective is not None:
        direct = [t for t in tensors if t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()]
        rest = [t for t in tensors if not (t.dtype == torch.float32 and t.is_cuda and t.is_contiguous())]
        work = [collective.broadcast(t.data, src=src) for t in direct]
        for w in work:
            w.wait()
        if rest and collective.world > 1:
            if not (dist.is_initialized() and dist.get_world_size() == collective.world):
                raise RuntimeError("broadcast_parameters: %d tensor(s) are not contiguous float32 device tensors "
                                   "(%s) and no torch.distributed group is up to carry them"
                                   % (len(rest), ", ".join(sorted({str(t.dtype) for t in rest}))))
            for t in rest:
                dist.broadcast(t.data, src=src)
        return
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t.data, src=src)
