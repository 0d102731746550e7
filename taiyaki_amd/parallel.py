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


def init_from_env(backend=None):
    """Rendezvous from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TK_FORCE_PROCESS_GROUP=1: build the (single-rank) RCCL communicator anyway -- lets a 1-GPU
    # box exercise the collective path, its streams and the NCCL watchdog thread
    forced = bool(os.environ.get("TK_FORCE_PROCESS_GROUP"))
    if (world > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
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


class FlatGradArena:
    """All trainable gradients as views into one contiguous fp32 buffer."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._always = dist.is_initialized() and bool(os.environ.get("TK_FORCE_PROCESS_GROUP"))
        self._work = None

    def zero(self):
        self.flat.zero_()

    def allreduce_async(self):
        """SUM over ranks (the reference's DDP averages; the 1/world factor is
        applied by `finish`)."""
        if self.world > 1 or self._always:
            self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
            self.flat.mul_(1.0 / self.world)


def broadcast_parameters(module, src=0):
    """Replaces the checkpoint-file + barrier handshake of the reference
    (bin/train_flipflop.py:380-392): rank 0's weights go out over RCCL."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
