"""The train step around the flip-flop loss: counterpart of `calculate_loss`
(bin/train_flipflop.py:145-198) and the optimiser step of `train_model`
(532-627), without per-step host synchronisation."""
import os

import torch

from taiyaki_amd import ctc, layers

# the fused (A) + (B) / nblk operator, plain CRF and cat-mod (TK_FUSED_LOSS=0: the two reference
# operators and autograd's add, as bin/train_flipflop.py:165-176 calls them)
FUSED_LOSS = os.environ.get("TK_FUSED_LOSS", "1") != "0"


def calculate_loss(net, indata, seqs, seqlens, sharpen=1.0, mod_cats=None,
                   can_mods_offsets=None, mod_cat_weights=None, ignore_empty=False):
    """lossvector = (A) crf / cat-mod loss + (B) logZ(outputs[:, :, :ntrans]) / nblk;
    loss = mean (bin/train_flipflop.py:161-182).

    `ignore_empty`: batches assembled on the device keep their shape when fewer chunks than
    asked for pass the filters and pad with zero-signal columns of sequence length 0
    (mapped_signal.sample_chunks); the reference trains on the shorter batch.  With the flag the
    mean runs over the columns with a sequence only (on the device, no sync), which is the same
    loss and the same gradients as the reference's shorter batch."""
    outputs = net(indata)
    nblk = float(outputs.shape[0])
    ntrans = outputs.shape[2]
    if FUSED_LOSS and outputs.is_cuda:
        # one operator, one gradient tensor, already d loss / d outputs (ctc.FlipFlopMeanLoss); cat-mod:
        # kernel B on the canonical columns first, folded into kernel A's writes
        weights = None
        if ignore_empty:
            live = (seqlens.to(outputs.device) > 0).to(torch.float32)
            weights = live / live.sum().clamp(min=1.0)
        return ctc.flipflop_mean_loss(outputs, seqs, seqlens, sharpen, weights, mod_cats, can_mods_offsets,
                                      mod_cat_weights)
    if mod_cats is not None:
        lossvector = ctc.cat_mod_flipflop_loss(outputs, seqs, seqlens, mod_cats,
                                               can_mods_offsets, mod_cat_weights, sharpen)
        ntrans -= ctc.n_mod_columns(can_mods_offsets)
    else:
        lossvector = ctc.crf_flipflop_loss(outputs, seqs, seqlens, sharpen)
    lossvector = lossvector + layers.flipflop_logpartition(outputs[:, :, :ntrans]) / nblk
    if ignore_empty:
        live = (seqlens.to(lossvector.device) > 0).to(lossvector.dtype)
        return (lossvector * live).sum() / live.sum().clamp(min=1.0), lossvector
    return lossvector.mean(), lossvector


class GateWatch:
    """Says so when the loss is losing time.  The CRF's linear-domain path hands a read it cannot
    represent (scores far outside the network's 5 tanh range, sharpening beyond 3.5, violent cat-mod
    logits, bands a few cells wide) to its log-domain kernel: a right answer at ~1000x the cost of
    the read (1 ms instead of 1 us at T = 800).  The kernels count those reads in the status word;
    every `every` steps this reads the count (non-strict mode: the one host sync it costs) and warns
    when more than `fraction` of the reads seen since the last check were redone."""

    def __init__(self, every=200, fraction=0.01):
        from taiyaki_amd import _lib
        self.every, self.fraction = every, fraction
        self.steps = self.reads = 0
        self.seen_total = _lib.gated_total()
        self.seen_retried = _lib.retried_total()
        self.last_fraction = 0.0
        self.last_retried_fraction = 0.0        # (round 6) reads swept again alone on the linear path: ~0.3 ms each, not 2.5

    def note(self, nreads):
        from taiyaki_amd import _lib
        self.steps += 1
        self.reads += int(nreads)
        if self.every <= 0 or self.steps % self.every or torch.cuda.is_current_stream_capturing():
            return
        if not _lib.is_strict():
            _lib.take_gate_count()
        total = _lib.gated_total()
        redone, self.seen_total = total - self.seen_total, total
        rt = _lib.retried_total()
        retried, self.seen_retried = rt - self.seen_retried, rt
        self.last_fraction = redone / max(1, self.reads)
        self.last_retried_fraction = retried / max(1, self.reads)
        if redone > self.fraction * self.reads:
            import warnings
            warnings.warn("flip-flop loss: %d of the last %d reads (%.1f %%) were redone by the log-domain kernel "
                          "(~1000x the cost of a read on the linear path): scores outside the range the linear path "
                          "represents (%d more were kept by the per-read retry) -- see taiyaki_amd.ctc.last_gate_count"
                          % (redone, self.reads, 100.0 * self.last_fraction, retried - redone if retried > redone else 0),
                          RuntimeWarning, stacklevel=3)
        self.reads = 0


class Trainer:
    """One optimiser step = forward, loss, backward, flat all-reduce, clip, AdamW.
    Defaults follow bin/_bin_argparse.py:16-193 (AdamW lr 4e-3, wd 0.01, eps 1e-6)."""

    def __init__(self, net, arena, lr=4e-3, weight_decay=0.01, eps=1e-6, grad_clip=None,
                 clip_num_mads=None, clip_window=1000):
        """`clip_num_mads` (reference default 0, `--gradient_clip_num_mads`; None = off) turns on
        the reference's adaptive clipping: per-parameter gradient maxima every step, clamp at
        median + num_mads * MAD of the last `clip_window` maxima once that many were seen
        (bin/train_flipflop.py:201-212, 575-578) -- on the device, see clipping.DeviceClipper.
        `grad_clip` is a fixed clip-by-value bound."""
        self.net = net
        self.arena = arena
        self.clipper = None
        if clip_num_mads is not None and arena.flat.is_cuda:
            from taiyaki_amd import clipping
            self.clipper = clipping.DeviceClipper(arena, clip_num_mads, clip_window)
        self.opt = torch.optim.AdamW(arena.params, lr=lr, weight_decay=weight_decay, eps=eps,
                                     betas=(0.9, 0.999),
                                     capturable=arena.flat.is_cuda)  # no host sync in step()
        self.grad_clip = grad_clip
        self.gate_watch = GateWatch()

    def step(self, batch):
        """`batch`: one batch (dict) or a list of sub-batches.  Sub-batches are the reference's way
        to fit a big batch (bin/train_flipflop.py:153-198): one forward / backward each, gradients
        accumulated and divided by their number (:190-193), ONE optimiser step; returns the mean of
        the sub-batch losses (:195).  The reference's DDP reduces after every sub-batch's backward;
        here only the last backward issues the (hook-overlapped) all-reduce -- the same average."""
        subs = batch if isinstance(batch, (list, tuple)) else [batch]
        self.arena.zero()
        hooks = self.arena.hooks_enabled
        total = None
        for k, sub in enumerate(subs):
            self.arena.hooks_enabled = hooks and k == len(subs) - 1
            loss, _ = calculate_loss(self.net, **sub)
            ctc.backward_unit(loss)
            total = loss.detach() if total is None else total + loss.detach()
        self.arena.hooks_enabled = hooks
        self.arena.allreduce_async()
        self.arena.finish(scale=1.0 / len(subs))
        self.clip()
        self.opt.step()
        self.gate_watch.note(sum(int(sub["seqlens"].numel()) for sub in subs))
        return loss if len(subs) == 1 else total / len(subs)

    def clip(self):
        """Gradient maxima / clipping between the all-reduce and the optimiser step."""
        if self.clipper is not None:
            if torch.cuda.is_current_stream_capturing():
                self.clipper.launch_kernels()       # whole-step capture: no host traffic inside
            else:
                self.clipper.step()
        if self.grad_clip is not None:
            self.arena.flat.clamp_(min=-self.grad_clip, max=self.grad_clip)


# Other threads of the process (the NCCL/RCCL watchdog polls events) must not invalidate a
# capture in progress: capture errors are scoped to the capturing thread.
_CAPTURE = dict(capture_error_mode="thread_local")


class GraphedTrainer:
    """The whole optimiser step captured ONCE into a hipGraph and replayed.

    The PyTorch-ROCm LSTM issues ~16,000 tiny per-timestep kernels per step at
    T = 800 x 5 layers; launched eagerly the step is bound by the host (190 ms),
    replayed from a graph it is bound by the GPU.  Requirements met here: static
    input buffers (each batch is copied in), sequence tensors resident on the
    device and padded to a fixed capacity, no host synchronisation inside the step
    (`_lib.set_strict(False)`, capturable AdamW), loss kernels launched on the
    capturing stream through the C ABI.
    """

    def __init__(self, trainer, example_batch, seq_capacity, max_seqlen=None):
        """`max_seqlen`: an upper bound on the sequence length of EVERY batch this trainer will see
        (the captured CRF launch is sized once: its waves per read cannot grow at replay).  None:
        unknown = the launch is sized for nblk + 1."""
        self.trainer = trainer
        dev = trainer.arena.flat.device
        self.max_seqlen = max_seqlen
        self.static = dict(
            indata=torch.zeros_like(example_batch["indata"], device=dev),
            seqs=torch.zeros(seq_capacity, dtype=torch.int32, device=dev),
            seqlens=torch.zeros_like(example_batch["seqlens"], dtype=torch.int32, device=dev))
        if example_batch.get("mod_cats") is not None:
            # cat-mod batches (bin/train_flipflop.py:116-142): per-position modification
            # categories next to the sequences; the two small tables stay constant
            self.static["mod_cats"] = torch.zeros(seq_capacity, dtype=torch.int32, device=dev)
            self.static["can_mods_offsets"] = example_batch["can_mods_offsets"]
            self.static["mod_cat_weights"] = example_batch["mod_cat_weights"]
        if example_batch.get("ignore_empty"):
            self.static["ignore_empty"] = True
        if max_seqlen is not None:
            ctc.set_max_seqlen(self.static["seqlens"], max_seqlen)
        self.loss = None
        self.graph = None

    def load(self, batch):
        """Copy a batch into the static buffers.  When the captured CRF launch was sized for
        `max_seqlen`, every batch must PROVE that it fits before it is copied in: by the hint its
        `seqlens` tensor carries (`ctc.set_max_seqlen`: bench.make_batches, mapped_signal.sample_chunks
        with `max_bases_per_chunk`), or -- `seqlens` on the host, as bin/train_flipflop.py:133-138
        builds them -- by looking.  A device tensor without a hint (it lost the attribute in a
        `.to()` / slice, or never had one) is refused: checking it would be a sync per step, and
        copying it in unchecked lets a longer read reach a launch that cannot hold it (the kernels
        then flag it and write NaN for that read -- loud, but a lost step)."""
        if self.max_seqlen is not None:
            sl = batch["seqlens"]
            hint = getattr(sl, "tk_max_seqlen", None)
            if hint is None:
                if torch.is_tensor(sl) and sl.is_cuda:
                    raise ValueError("this step was captured for sequences of at most %d bases: a `seqlens` tensor on "
                                     "the device must carry its maximum (ctc.set_max_seqlen(seqlens, max)) -- keep it "
                                     "on the host or set the hint where the batch is assembled" % self.max_seqlen)
                hint = int(sl.max()) if len(sl) else 0
            if hint > self.max_seqlen:
                raise ValueError("batch with sequences up to %d bases for a step captured for at most %d"
                                 % (hint, self.max_seqlen))
        self.static["indata"].copy_(batch["indata"], non_blocking=True)
        n = batch["seqs"].numel()
        self.static["seqs"][:n].copy_(batch["seqs"], non_blocking=True)
        self.static["seqlens"].copy_(batch["seqlens"], non_blocking=True)
        if "mod_cats" in self.static:
            self.static["mod_cats"][:n].copy_(batch["mod_cats"], non_blocking=True)

    def capture(self, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.trainer.step(self.static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        arena = self.trainer.arena
        hooks = arena.hooks_enabled
        arena.hooks_enabled = False         # one captured all-reduce, not hook-issued slices
        try:
            with torch.cuda.graph(self.graph, **_CAPTURE):
                self.loss = self.trainer.step(self.static)
        finally:
            arena.hooks_enabled = hooks
        torch.cuda.synchronize()

    def step(self, batch):
        """Replay.  The clipper's host half runs around the replay exactly as in the eager step:
        last step's maxima -> rolling statistics -> thresholds (uploaded before the replay, so the
        captured clamp kernel sees them), and this step's maxima start their way to the host."""
        clipper = self.trainer.clipper
        self.load(batch)
        if clipper is not None:
            clipper.collect()
        self.graph.replay()
        if clipper is not None:
            clipper.copy_maxima_async()
        self.trainer.gate_watch.note(batch["seqlens"].numel())
        return self.loss


def _aliased_adamw(trainer):
    """The optimiser of the hybrid trainers.  The captured autograd graph saved (views of) the
    parameters; an in-place optimiser update of the parameters themselves would trip autograd's
    version check on the next eager backward.  The optimiser therefore updates ALIASES of the
    parameters (same storage, own version counters) -- the replayed forward and the eager backward of
    one step always see the same, current weights.  The learning rate lives in a DEVICE scalar
    (capturable AdamW reads it at replay time): `_set_lr` between replays is what a per-iteration
    schedule needs.  Returns (aliases, optimiser)."""
    arena = trainer.arena
    alias = []
    for p in arena.params:
        a = p.data.requires_grad_(True)
        a.grad = p.grad
        alias.append(a)
    g = trainer.opt.param_groups[0]
    lr = g["lr"]
    lr = lr.detach().clone().to(arena.flat.device) if torch.is_tensor(lr) else \
        torch.tensor(float(lr), dtype=torch.float32, device=arena.flat.device)
    opt = torch.optim.AdamW(alias, lr=lr, weight_decay=g["weight_decay"], eps=g["eps"], betas=g["betas"],
                            capturable=True)
    return alias, opt


def _set_lr(opt, lr):
    for g in opt.param_groups:
        if torch.is_tensor(g["lr"]):
            g["lr"].fill_(float(lr))
        else:
            g["lr"] = float(lr)


class HybridGraphTrainer(GraphedTrainer):
    """Forward + loss replayed from a hipGraph, backward launched eagerly, AdamW replayed.

    MIOpen's fused LSTM is the fastest RNN PyTorch-ROCm offers, but its BACKWARD cannot be
    captured on ROCm 7.2 (hipBLASLt refuses a call inside stream capture and aborts).  Its
    forward can: so the forward half of the step (conv, 5 LSTM layers = ~8,000 per-timestep
    launches, the HIP flip-flop loss kernels) is captured once together with the autograd
    graph it builds; every step replays it into the same buffers and runs
    `loss.backward(retain_graph=True)` through that autograd graph eagerly.  The host then
    only issues the backward launches -- about half of the step's 16,000.
    """

    def __init__(self, trainer, example_batch, seq_capacity, max_seqlen=None):
        super().__init__(trainer, example_batch, seq_capacity, max_seqlen)
        self.alias, self.opt = _aliased_adamw(trainer)

    def set_lr(self, lr):
        """The reference steps its learning-rate schedule every iteration
        (bin/train_flipflop.py:605-607); the replayed optimiser graph reads the rate from a device
        scalar, so a new value is one tiny fill between replays."""
        _set_lr(self.opt, lr)

    def _eager_step(self):
        tr = self.trainer
        tr.arena.zero()
        loss, _ = calculate_loss(tr.net, **self.static)
        ctc.backward_unit(loss)
        tr.arena.allreduce_async()
        tr.arena.finish()
        self._clip_and_step()

    def capture(self, warmup=3):
        tr = self.trainer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, **_CAPTURE):
            tr.arena.zero()
            self.loss, _ = calculate_loss(tr.net, **self.static)
        torch.cuda.synchronize()
        # capture only records: replay once so that the loss and the saved activations exist,
        # finish that step eagerly, then capture the optimiser step
        self.graph.replay()
        self._tail_eager()
        torch.cuda.synchronize()
        self.opt_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.opt_graph, **_CAPTURE):
            self.opt.step()
        torch.cuda.synchronize()

    def _clip_and_step(self):
        self.trainer.clip()
        self.opt.step()

    def _tail_eager(self):
        ctc.backward_unit(self.loss, retain_graph=True)
        self.trainer.arena.allreduce_async()
        self.trainer.arena.finish()
        self._clip_and_step()

    def step(self, batch):
        self.load(batch)
        self.graph.replay()
        ctc.backward_unit(self.loss, retain_graph=True)
        self.trainer.arena.allreduce_async()
        self.trainer.arena.finish()
        self.trainer.clip()         # eager: two tiny launches + an async copy of the maxima
        self.opt_graph.replay()
        self.trainer.gate_watch.note(batch["seqlens"].numel())
        return self.loss


class GraphCacheTrainer:
    """The hybrid scheme (`HybridGraphTrainer`) for the reference's own schedule, which draws a new chunk length for
    every iteration and rescales the batch with it (bin/train_flipflop.py:554-563:
    batch_chunk_len in [chunk_len_min, chunk_len_max], sub_batch_size = min_sub_batch_size *
    chunk_len_max / batch_chunk_len): one captured forward + loss graph PER SHAPE
    (indata.shape = (chunk_len, nbatch, 1)), captured the first time the shape is seen and
    replayed from then on; the optimiser state and its graph are shared.  The caller keeps the
    number of shapes small by drawing chunk lengths from a grid (`bucket_chunk_len`).

    Capturing a new shape costs one eager forward / backward for the allocator and does NOT
    touch the weights: the step that follows is this batch's ordinary step."""

    def __init__(self, trainer, seq_capacity_per_chunk, max_seqlen_of=None, max_graphs=32):
        """`seq_capacity_per_chunk(chunk_len) -> int`: bases reserved per chunk of that length;
        `max_seqlen_of(chunk_len) -> int or None`: bound on a sequence's length (sizes the CRF
        launch of that shape; None = unknown)."""
        self.trainer = trainer
        self.seq_capacity_per_chunk = seq_capacity_per_chunk
        self.max_seqlen_of = max_seqlen_of or (lambda chunk_len: None)
        self.max_graphs = max_graphs
        self.entries = {}
        self.opt_graph = None
        self.nsteps = 0
        self.hits = self.misses = 0         # steps that replayed a shape's graph / that had to capture it first
        self.alias, self.opt = _aliased_adamw(trainer)

    def set_lr(self, lr):
        _set_lr(self.opt, lr)

    @staticmethod
    def key_of(batch):
        return tuple(batch["indata"].shape) + (batch.get("mod_cats") is not None,)

    def _capture_shape(self, batch):
        if len(self.entries) >= self.max_graphs:
            raise RuntimeError("GraphCacheTrainer: more than %d distinct batch shapes -- draw chunk lengths "
                               "from a grid (bucket_chunk_len)" % self.max_graphs)
        chunk_len, nbatch = batch["indata"].shape[0], batch["indata"].shape[1]
        one = GraphedTrainer(self.trainer, batch, seq_capacity=nbatch * self.seq_capacity_per_chunk(chunk_len),
                             max_seqlen=self.max_seqlen_of(chunk_len))
        one.load(batch)
        tr = self.trainer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # allocator / autotuner warm-up at this shape: forward + loss + backward, no optimiser
            # step (the gradient arena is zeroed again by the captured graph)
            for _ in range(2):
                tr.arena.zero()
                hooks = tr.arena.hooks_enabled
                tr.arena.hooks_enabled = False
                try:
                    loss, _ = calculate_loss(tr.net, **one.static)
                    ctc.backward_unit(loss)
                finally:
                    tr.arena.hooks_enabled = hooks
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        one.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(one.graph, **_CAPTURE):
            tr.arena.zero()
            one.loss, _ = calculate_loss(tr.net, **one.static)
        torch.cuda.synchronize()
        return one

    def step(self, batch):
        key = self.key_of(batch)
        one = self.entries.get(key)
        if one is None:
            self.misses += 1
            one = self.entries[key] = self._capture_shape(batch)
        else:
            self.hits += 1
        one.load(batch)
        one.graph.replay()
        ctc.backward_unit(one.loss, retain_graph=True)
        self.trainer.arena.allreduce_async()
        self.trainer.arena.finish()
        self.trainer.clip()
        if self.opt_graph is not None:
            self.opt_graph.replay()
        else:
            self.opt.step()             # the first step creates the optimiser state eagerly ...
            torch.cuda.synchronize()
            self.opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.opt_graph, **_CAPTURE):
                self.opt.step()         # ... (capture records, it does not run) replayed from the second on
            torch.cuda.synchronize()
        self.nsteps += 1
        self.trainer.gate_watch.note(batch["seqlens"].numel())
        return one.loss


def bucket_chunk_len(chunk_len, stride, bucket_blocks=100):
    """The reference forces a drawn chunk length to a multiple of the stride
    (bin/train_flipflop.py:554-557); a trainer that replays captured graphs wants few distinct
    lengths: round DOWN to a multiple of bucket_blocks strides (500 samples at stride 5: eleven
    lengths over the default 3000-8000 range)."""
    q = stride * bucket_blocks
    return max(q, (int(chunk_len) // q) * q)
