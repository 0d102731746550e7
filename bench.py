#!/usr/bin/env python
"""bench.py -- flip-flop train-step throughput + loss-kernel rooflines on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,3,4,5}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` (N > 1) without a torchrun environment LAUNCHES the N ranks itself (the same
torch.distributed.run command, one rank per GPU, like the reference's workflow/test_multiGPU.sh:47-55)
and fails loudly when fewer than N devices are visible.

One "step" = one optimiser step of the reference's flip-flop trainer
(bin/train_flipflop.py:544-622).  `--config` selects the BASELINE.json configuration in SURVEY.md
section 8's numbering (BASELINE.json configs[k-1]); the default 2 is the one the metric is quoted on:
  1  mGru_flipflop size 96 stride 2, chunk_len 2000 (T = 1000), 64 chunks   (configs[0], the
     train_abinitio.py plumbing case; here on the GPU with the HIP loss)
  2  mLstm_flipflop size 256 stride 5, chunk_len 4000 (T = 800), 128 chunks per GPU  (configs[1])
  3  = 2 with --gpus 8: global batch 1024, RCCL all-reduce over xGMI                  (configs[2])
  4  mLstm_cat_mod_flipflop (5mC + 6mA, 46 outputs), cat-mod loss, mod_factor 8       (configs[3])
  5  mLstm_flipflop, chunk_len 8000 (T = 1600), 64 chunks per GPU                     (configs[4])
Conv/RNN stack in PyTorch-ROCm fp32 -> HIP flip-flop CRF (or cat-mod) loss + HIP logZ -> backward
-> bucketed RCCL all-reduce overlapped with backward -> gradient maxima / clipping -> AdamW.
Forward + loss and AdamW are replayed from hipGraphs, the MIOpen RNN backward (not capturable) is
launched eagerly.  Synthetic chunks (resident in HBM before the timed region), random-init weights.
Weak scaling: per-GPU batch is fixed, value = all ranks' chunks / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline"        the logZ forward-backward op on the tensor BASELINE.json's north_star names
                    for the roofline target (T=4000 blocks, N=256 reads): HIP events on the
                    launching stream; achieved = 3*T*N*S*4 bytes / mean duration;
                    traffic = HBM bytes from rocprofv3 PMC passes (measured in this run when
                    rocprofv3 is on PATH, else the committed profile with its kernel hash)
  "roofline_in_step" the same op at the shape the train step itself launches
  "roofline_crf"    kernel A (sequence CRF, sweep + posterior) at the step's shape with realistic
                    sequence lengths and at the north_star shape
  "loss_path"       the whole loss path (A + B) in chunks/s on the GPU, on the host cores, and
                    on the host cores including the D->H / H->D copies of the score and gradient
                    tensors the reference design incurs (ctc.pyx:119, 139-141)
  "rccl"            (N > 1) ranks, bytes and event-timed duration of the gradient all-reduce, whole and
                    slice by slice (`bucket_us`); "per_rank_ms": every rank's own ms/step (min / max /
                    all); "cores_per_rank": the host cores each rank is pinned to
  "cpu_baseline"    the reference C (oracle/_ref) or the oracle port on host cores.
"""
import argparse
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import time

# hipBLASLt's bias-epilogue path copies its user arguments with a call that is illegal
# during stream capture (hard abort); plain rocBLAS GEMMs capture fine.
os.environ.setdefault("DISABLE_ADDMM_CUDA_LT", "1")
os.environ.setdefault("TORCH_BLAS_PREFER_HIPBLASLT", "0")
os.environ.setdefault("ROCBLAS_USE_HIPBLASLT", "0")            # rocBLAS -> its own Tensile kernels
os.environ.setdefault("MIOPEN_GEMM_ENFORCE_BACKEND", "1")      # MIOpen RNN GEMMs -> rocBLAS
# HIP spreads streams and hipGraph branches over several hardware queues and pays a
# cross-queue signal (~10 us) whenever the ~16,000 tiny dependent launches of a step hop
# between them.  One queue keeps the whole step in order on the hardware: 167 -> 110 ms per
# step on one GPU (124-127 ms with two queues).  The limit is per PRIORITY level: RCCL's
# collectives run on a high-priority stream (taiyaki_amd/parallel.py) and therefore in a
# hardware queue of their own, so multi-GPU runs keep the single compute queue too
# (tools/queue_probe.py shows a high-priority stream running beside a busy normal one at
# GPU_MAX_HW_QUEUES=1).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)

# SURVEY.md section 8 table (model, chunk_len, stride, chunks per GPU, size, samples per base)
CONFIGS = {
    1: dict(model="mGru_flipflop", chunk_len=2000, stride=2, batch=64, size=96, spb=9.0,
            label="configs[0]: mGru_flipflop r9 DNA (train_abinitio.py defaults)"),
    2: dict(model="mLstm_flipflop", chunk_len=4000, stride=5, batch=128, size=256, spb=9.0,
            label="configs[1]: mLstm_flipflop r9.4.1 DNA"),
    3: dict(model="mLstm_flipflop", chunk_len=4000, stride=5, batch=128, size=256, spb=9.0,
            label="configs[2]: mLstm_flipflop r9.4.1 DNA, global batch 1024 over 8 GPUs"),
    4: dict(model="mLstm_cat_mod_flipflop", chunk_len=4000, stride=5, batch=128, size=256, spb=9.0,
            label="configs[3]: mLstm_cat_mod_flipflop (5mC+6mA) r9.4.1, cat-mod loss"),
    5: dict(model="mLstm_flipflop", chunk_len=8000, stride=5, batch=64, size=256, spb=9.5,
            label="configs[4]: mLstm_flipflop r10.3 DNA, long chunks"),
}
CAN_NMODS = (1, 1, 0, 0)        # alphabet ACGTZY: 6mA on A, 5mC on C (layers.py:1441-1460)
MOD_FACTOR = 8.0                # --mod_factor start value (bin/_bin_argparse.py:178)
PATH_BUFFER = 1.1               # --filter_path_buffer default (bin/_bin_argparse.py)


def make_batches(nbatch, chunk_len, stride, seed, dev, n=4, spb=9.0, cat_mod=False):
    from taiyaki_amd import synth
    T = chunk_len // stride
    out = []
    for i in range(n):
        s = seed * 1000 + i
        seqlens = synth.realistic_seqlens(T, nbatch, s, chunk_len, spb)
        seqs, bases = synth.sequences(seqlens, s)
        sig = synth.signal_chunks(chunk_len, nbatch, s)
        b = dict(indata=torch.from_numpy(sig).to(dev),
                 seqs=torch.from_numpy(seqs).to(device=dev, dtype=torch.int32),
                 seqlens=torch.from_numpy(seqlens).to(device=dev, dtype=torch.int32))
        # the lengths are known on the host when a batch is assembled (bin/train_flipflop.py:133-138):
        # the device tensor carries its maximum along, the CRF launch is sized by it without a sync
        from taiyaki_amd import ctc
        ctc.set_max_seqlen(b["seqlens"], int(seqlens.max()), bulk=ctc.bulk_of(seqlens))
        if cat_mod:
            b["mod_cats"] = torch.from_numpy(synth.mod_cats(bases, s, CAN_NMODS)).to(device=dev, dtype=torch.int32)
            b["can_mods_offsets"] = synth.can_mods_offsets(CAN_NMODS)
            # train_flipflop.py:167-170, 312-313: mod_cat_weights = ones, times the mod_factor schedule
            b["mod_cat_weights"] = np.full(len(CAN_NMODS) + int(sum(CAN_NMODS)), MOD_FACTOR, dtype=np.float32)
        out.append(b)
    return out


# ---------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`--gpus N` outside torchrun: start the N ranks ourselves, one per GPU, over RCCL (the
    reference launches its multi-GPU run the same way, workflow/test_multiGPU.sh:47-55,
    bin/train_flipflop.py:255-268).  The JSON line of rank 0 is this process's output."""
    if not args.dry_launch and not os.environ.get("TK_BENCH_SHARE_GPU"):
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible -- refusing to measure fewer "
                             "ranks than asked for" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    pr = subprocess.run(cmd, env=env)
    raise SystemExit(pr.returncode)


# ---------------------------------------------------------------------------------------------
# kernel timing
# ---------------------------------------------------------------------------------------------
def _events_mean_min(fn, reps, warm=20):
    """Mean / min duration (s) of `fn` via HIP events on the launching (current) stream."""
    for _ in range(warm):       # steady state: clocks and caches settle over the first ~15 launches
        fn()
    torch.cuda.synchronize()
    # Keep the GPU busy while the host enqueues the timed launches: the ~40 us of Python between
    # an event record and the first kernel launch must not show up as GPU idle time inside the
    # event window (the events then bracket exactly the op's kernels, back to back).
    # (8e6 cycles per ten launches: a slow or busy host -- a box whose pinned copies ran at a sixteenth of the
    # usual rate was seen to take 5 ms for 50 enqueues, longer than the 2e6 per ten this used to wait, and the
    # op's mean came out 5 us above the sum of its kernels' durations under rocprofv3)
    torch.cuda._sleep(int(8.0e6 * max(1, reps // 10)))
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return float(np.mean(ms)) * 1e-3, float(ms[0]) * 1e-3


class LossOps:
    """The loss path's two operators on synthetic tensors resident in HBM, launched through the
    C ABI with every device buffer allocated ONCE (what a captured train step replays): the timed
    region contains the kernels only."""

    def __init__(self, T, N, dev, realistic_chunk_len=None, spb=9.0, seed=1, cat_mod=False):
        from taiyaki_amd import _lib, synth
        self.T, self.N, self.dev = T, N, dev
        seqlens = None
        if realistic_chunk_len:
            seqlens = synth.realistic_seqlens(T, N, 17000 + seed, realistic_chunk_len, spb)
        inp = synth.crf_case(T, N, seed, seqlens=seqlens, nmods_per_base=CAN_NMODS if cat_mod else None)
        if cat_mod:
            synth.normalise_mod_columns(inp)    # log-softmax mod columns, like the producer layer's
        self.S = inp["scores"].shape[2]
        self.x = torch.from_numpy(inp["scores"]).to(dev)
        self.x40 = self.x[:, :, :40].contiguous() if cat_mod else self.x
        self.seqs = torch.from_numpy(inp["seqs"]).to(device=dev, dtype=torch.int32)
        self.seqlens = torch.from_numpy(inp["seqlens"]).to(device=dev, dtype=torch.int32)
        self.maxlen = int(inp["seqlens"].max())
        from taiyaki_amd import ctc as _ctc
        self.bulk = _ctc.bulk_of(inp["seqlens"])       # (tk_seq_labels.bulk_seqlen: what the operators pass for host lengths)
        self.mod = None
        if cat_mod:
            self.mod = (torch.from_numpy(inp["mod_cats"]).to(device=dev, dtype=torch.int32),
                        inp["can_mods_offsets"], inp["mod_cat_weights"])
        self.host = inp
        L = _lib.lib()
        total = self.seqs.numel()
        self.seqoff = torch.empty(N + 1, dtype=torch.int64, device=dev)
        self.stay = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        self.move = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
        self.modidx = self.modfact = self.cmo = self.mcw = None
        if cat_mod:
            self.modidx = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
            self.modfact = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
            self.cmo = torch.as_tensor(inp["can_mods_offsets"], dtype=torch.int32).to(dev)
            self.mcw = torch.as_tensor(inp["mod_cat_weights"], dtype=torch.float32).to(dev)
        self.cost = torch.empty(N, dtype=torch.float32, device=dev)
        self.grad = torch.empty_like(self.x)
        self.crf_wsb = L.tk_crf_flipflop_workspace_bytes(self.S, T, N, self.maxlen, 1)
        self.crf_ws = torch.empty(self.crf_wsb, dtype=torch.uint8, device=dev)
        self.logz = torch.empty(N, dtype=torch.float32, device=dev)
        self.lgrad = torch.empty_like(self.x40)
        self.lz_wsb = L.tk_flipflop_logz_workspace_bytes(T, N, 4)
        self.lz_ws = torch.empty(self.lz_wsb, dtype=torch.uint8, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.aux_bytes = L.tk_flipflop_loss_fused_aux_bytes(T, N, 4, self.S)
        self.aux = torch.empty(self.aux_bytes, dtype=torch.uint8, device=dev) if self.aux_bytes else None
        # (round 5: the index build rides in kernel A's first launch; True = round 4's two calls, for A/B)
        self.separate_index_build = False

    def labels(self):
        """tk_seq_labels of these inputs (the entry points that build their indices inside their first launch)."""
        import ctypes
        from taiyaki_amd import _lib
        p = _lib.ptr
        self._labels = _lib.SeqLabels(p(self.seqs), self.seqs.numel(), 4, p(self.mod[0]) if self.mod else None,
                                      p(self.cmo), p(self.mcw), self.bulk)
        return ctypes.byref(self._labels)

    def crf(self):
        from taiyaki_amd import _lib
        L, p = _lib.lib(), _lib.ptr
        st = _lib.stream_ptr()
        if not self.separate_index_build:
            rc = L.tk_crf_flipflop_labels_dev(p(self.x), self.S, self.T, self.N, self.labels(), p(self.seqlens), p(self.seqoff),
                                              p(self.stay), p(self.move), p(self.modidx), p(self.modfact), self.maxlen, 40,
                                              1.0, 1.0, 1.0, p(self.cost), p(self.grad), p(self.crf_ws), self.crf_wsb,
                                              p(self.status), st)
            _lib.check(rc, "tk_crf_flipflop_labels_dev")
            return
        rc = L.tk_flipflop_build_indices_dev(p(self.seqs), p(self.seqlens), self.N, self.seqs.numel(), 4,
                                             p(self.mod[0]) if self.mod else None, p(self.cmo), p(self.mcw),
                                             p(self.seqoff), p(self.stay), p(self.move), p(self.modidx),
                                             p(self.modfact), p(self.status), st)
        _lib.check(rc, "tk_flipflop_build_indices_dev")
        rc = L.tk_crf_flipflop_dev(p(self.x), self.S, self.T, self.N, p(self.stay), p(self.move), p(self.modidx),
                                   p(self.modfact), p(self.seqlens), p(self.seqoff), self.maxlen, 40, 1.0, 1.0, 1.0,
                                   p(self.cost), p(self.grad), p(self.crf_ws), self.crf_wsb, p(self.status), st,
                                   p(self.mcw) if self.mod else None)
        _lib.check(rc, "tk_crf_flipflop_dev")

    def logz_op(self):
        from taiyaki_amd import _lib
        L, p = _lib.lib(), _lib.ptr
        rc = L.tk_flipflop_logz_dev(p(self.x40), self.T, self.N, 4, p(self.logz), p(self.lgrad), p(self.lz_ws),
                                    self.lz_wsb, p(self.status), _lib.stream_ptr())
        _lib.check(rc, "tk_flipflop_logz_dev")

    def both(self):
        """The loss path as the train step launches it: the fused (A) + (B) / nblk entry point
        (plain CRF and cat-mod)."""
        from taiyaki_amd import _lib
        L, p = _lib.lib(), _lib.ptr
        st = _lib.stream_ptr()
        if not self.separate_index_build:
            rc = L.tk_flipflop_loss_fused_labels_dev(p(self.x), self.T, self.N, self.S, self.labels(), p(self.seqlens),
                                                     p(self.seqoff), p(self.stay), p(self.move), p(self.modidx),
                                                     p(self.modfact), self.maxlen, 1.0, 1.0 / self.N, None, p(self.cost),
                                                     p(self.grad), p(self.logz), p(self.crf_ws), self.crf_wsb, p(self.lz_ws),
                                                     self.lz_wsb, p(self.aux), self.aux_bytes, p(self.status), st)
            _lib.check(rc, "tk_flipflop_loss_fused_labels_dev")
            return
        rc = L.tk_flipflop_build_indices_dev(p(self.seqs), p(self.seqlens), self.N, self.seqs.numel(), 4,
                                             p(self.mod[0]) if self.mod else None, p(self.cmo), p(self.mcw),
                                             p(self.seqoff), p(self.stay), p(self.move), p(self.modidx),
                                             p(self.modfact), p(self.status), st)
        _lib.check(rc, "tk_flipflop_build_indices_dev")
        rc = L.tk_flipflop_loss_fused_dev(p(self.x), self.T, self.N, 4, self.S, p(self.stay), p(self.move),
                                          p(self.modidx), p(self.modfact), p(self.seqlens),
                                          p(self.seqoff), self.maxlen, 1.0, 1.0 / self.N, None, p(self.cost), p(self.grad),
                                          p(self.logz), p(self.crf_ws), self.crf_wsb, p(self.lz_ws), self.lz_wsb,
                                          p(self.aux), self.aux_bytes, p(self.status), st,
                                          p(self.mcw) if self.mod else None)
        _lib.check(rc, "tk_flipflop_loss_fused_dev")

    def finite(self):
        return (int(self.status.item()) & 0xff) == 0

    def gated_reads(self):
        """Reads the linear path handed to the log-domain kernel in ONE call of the CRF op on these inputs (the
        status word's count, include/taiyaki_amd_flipflop.h: TK_STATUS_GATED_SHIFT)."""
        return self.gate_counts()[0]

    def gate_counts(self):
        """(reads redone in the log domain, reads retried alone on the linear path) in ONE call of the CRF op on these
        inputs: the status word's two counts (TK_STATUS_GATED_SHIFT, TK_STATUS_RETRIED_SHIFT)."""
        torch.cuda.synchronize()
        self.status.zero_()
        self.crf()
        bits = int(self.status.item()) & 0xffffffff
        return (bits >> 8) & 0xfff, (bits >> 20) & 0xfff


def kernel_hash():
    """Hash of the kernel sources: ties a committed PMC profile to the code it was taken from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "taiyaki_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def committed_traffic(op, T, N):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary for (op, T, N)."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_traffic.json") and "_pmc_" in name:
            with open(os.path.join(pdir, name)) as fh:
                d = json.load(fh)
            if d.get("op", "logz") == op and d["shape"]["T"] == T and d["shape"]["N"] == N:
                best = (name, d)
    if best is None:
        return None, None
    name, d = best
    src = dict(measured="profiles/" + name, kernel_hash=d.get("kernel_hash"))
    src["current"] = d.get("kernel_hash") == kernel_hash()
    return d["traffic_bytes"], src


CLOCK_HZ = 2.4e9            # MI355X_MICROARCH.md: peak engine clock
N_SIMD = 1024               # 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles


def issue_floor(op, T, N, realistic, W, BK):
    """Kernel A is bound by instruction issue, not by HBM: its floor is stated in those terms, from the
    shader-sequencer counters of the committed profile (tools/sq_counters.py, same kernel hash):
      sweep_valu_us      the sweep's VALU wave-instructions x 4 cycles / (1024 SIMDs x clock): the chip's VALU
                         issue rate, perfectly balanced
      sweep_chain_us     the recurrence itself: (NB + W - 1) phases x BK steps x 13 cycles (fma -> two wait states
                         -> DPP fmac, one dependent pair per step at the measured 5.4 cycles per instruction)
      posterior_valu_us  the gradient pass's VALU instructions at the chip's rate
    issue_floor_us = max(sweep floors) + the gradient pass's (the two launches are serial).  Returns None when
    no counters are committed for this shape."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_sq_counters.json"):
            with open(os.path.join(pdir, name)) as fh:
                d = json.load(fh)
            rec = d.get("shapes", {}).get("%s:%d:%d:%d" % (op, T, N, realistic))
            if rec and "sweep" in rec and "posterior" in rec:
                best = (name, d, rec)
    if best is None:
        return None
    name, d, rec = best
    valu = lambda r: rec[r].get("SQ_INSTS_VALU", 0.0) * 4.0 / (N_SIMD * CLOCK_HZ) * 1e6      # noqa: E731
    NB = (T + BK - 1) // BK
    chain = (NB + W - 1) * BK * 13.0 / CLOCK_HZ * 1e6
    out = dict(sweep_valu_us=round(valu("sweep"), 2), sweep_chain_us=round(chain, 2),
               posterior_valu_us=round(valu("posterior"), 2),
               waves_issuing_frac=round(rec["sweep"].get("SQ_ACTIVE_INST_ANY", 0.0) / max(1.0, rec["sweep"].get("SQ_WAVE_CYCLES", 1.0)), 3),
               source="profiles/" + name, kernel_hash=d.get("kernel_hash"), current=d.get("kernel_hash") == kernel_hash(),
               phases=NB + W - 1, block_steps=BK)
    out["issue_floor_us"] = round(max(out["sweep_valu_us"], out["sweep_chain_us"]) + out["posterior_valu_us"], 2)
    return out


def measure_traffic_now(specs, timeout=240):
    """FETCH_SIZE / WRITE_SIZE of the listed (op, T, N, realistic) launches, measured in this
    run: two rocprofv3 --pmc passes (the counters do not fit one pass) over tools/pmc_traffic.py.
    Returns {key: bytes} or {} when rocprofv3 is missing / fails / times out."""
    if shutil.which("rocprofv3") is None or os.environ.get("TK_BENCH_NO_PMC"):
        return {}
    tool = os.path.join(ROOT, "tools", "pmc_traffic.py")
    arg = ",".join("%s:%d:%d:%d" % s for s in specs)
    try:
        pr = subprocess.run([sys.executable, tool, "--ops", arg, "--json"], capture_output=True, text=True,
                            timeout=timeout, env={k: v for k, v in os.environ.items()
                                                  if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if pr.returncode == 0 and line else {}
    except Exception:
        return {}


def roofline_record(kernel, alg, mean_s, min_s, reps, traffic, source):
    return dict(bound="hbm", kernel=kernel, achieved=round(alg / mean_s / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(alg / mean_s / 1e9 / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=source,
                algorithmic_bytes=alg, mean_us=round(mean_s * 1e6, 2), min_us=round(min_s * 1e6, 2), launches=reps)


# ---------------------------------------------------------------------------------------------
# CPU legs (the only part of this file that touches oracle/)
# ---------------------------------------------------------------------------------------------
def cpu_baseline(inp, budget_s=12.0):
    """The loss path (A: crf / cat-mod grad, B: logZ fwd-bwd) on the host cores, on the SAME host
    arrays `loss_path.gpu_ms` is measured on (`LossOps.host`: the step's shape, realistic sequence
    lengths, same seed).  A runs on the genuine reference C when oracle/_ref was built (kind =
    "reference"), else on the oracle port.  BASELINE.md section 3's protocol: one warm-up, then the
    MEDIAN of >= 5 repetitions (as many as fit the time budget)."""
    import oracle
    oracle.build()
    cores = os.cpu_count() or 1
    use_ref = oracle.ref_available()
    T, N, S = inp["scores"].shape
    cat_mod = "mod_cats" in inp
    sc40 = np.ascontiguousarray(inp["scores"][:, :, :40]) if cat_mod else inp["scores"]

    def once():
        if cat_mod:
            oracle.cat_mod_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], inp["mod_cats"],
                                         inp["can_mods_offsets"], inp["mod_cat_weights"], 1.0, use_ref=use_ref)
        else:
            oracle.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0, use_ref=use_ref)
        oracle.flipflop_logz_grad(sc40)

    def run(threads, budget):
        oracle.set_threads(threads)
        once()                                          # warm-up
        times, t_start = [], time.perf_counter()
        while len(times) < 5 or time.perf_counter() - t_start < budget:
            t0 = time.perf_counter()
            once()
            times.append(time.perf_counter() - t0)
        return times, time.perf_counter() - t_start

    threads = min(cores, 8)     # the reference's own advice: OMP_NUM_THREADS=8 (README.md:362-372)
    times, el = run(threads, budget_s)
    med = float(np.median(times))
    out = dict(value=round(N / med, 2),
               unit="chunks/s (loss path only: %s grad + logZ fwd-bwd; compare with loss_path.gpu_chunks_per_s, "
                    "not with value)" % ("cat-mod" if cat_mod else "crf"),
               cores=threads, kind="reference" if use_ref else "port",
               sample="median of %d reps (after 1 warm-up, %.1f s) of T=%d N=%d S=%d, the SAME host arrays as "
                      "loss_path.gpu_ms (the step's shape, realistic sequence lengths up to %d); host has %d "
                      "cores; A = %s, B = oracle port (the reference's B is a T-step torch loop, layers.py:1277-1299)"
                      % (len(times), el, T, N, S, int(np.max(inp["seqlens"])), cores,
                         "genuine reference C (oracle/_ref)" if use_ref else "oracle port"),
               median_ms=round(med * 1e3, 3), min_ms=round(min(times) * 1e3, 3), reps=len(times))
    if cores > threads:
        allc = min(cores, N)
        times2, _ = run(allc, budget_s / 2)
        out["value_all_cores"] = round(N / float(np.median(times2)), 2)
        out["cores_all"] = allc
        out["cores_all_note"] = ("min(host cores = %d, reads in the batch = %d) threads: the reference parallelises over reads "
                                 "(c_crf_flipflop.c:453, one read per OpenMP iteration), so more threads than reads would idle; "
                                 "on this host that is SLOWER than 8 threads (dynamic scheduling of %d short tasks over %d threads "
                                 "that span sockets) -- `value` is the reference's own recommended setting" % (cores, N, N, allc))
    return out


def reference_copies_ms(T, N, S, dev, reps=10):
    """The device<->host traffic the reference design adds to every loss call: logprob D->H
    (ctc.pyx:119) and the gradient H->D (ctc.pyx:139-141, pinned like the reference's)."""
    x = torch.empty(T, N, S, device=dev)
    h = torch.empty(T, N, S).pin_memory()
    for _ in range(2):
        h.copy_(x)
        x.copy_(h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        h.copy_(x)
        x.copy_(h)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def forced_group_overhead(args, argv, plain_ms):
    """N = 1 only: what the data-parallel machinery costs before any wire is involved.  A child runs
    the same step with a forced ONE-rank RCCL process group (TK_FORCE_PROCESS_GROUP=1: communicator,
    watchdog, the arena's all-reduce and its stream joins, the 1 / world pass) while this process
    idles; overhead_ms = its ms/step - the plain step's.  The part of north_star's 1 -> 8 target
    that a one-GPU box can measure."""
    keep = []
    skip = 0
    for a in argv:
        if skip:
            skip -= 1
        elif a in ("--gpus",):
            skip = 1
        elif a not in ("--no-cpu-baseline", "--no-pmc", "--no-rowk", "--no-kernel-records"):
            keep.append(a)
    cmd = [sys.executable, os.path.abspath(__file__)] + keep + ["--gpus", "1", "--no-kernel-records",
                                                               "--steps", str(args.steps), "--warmup", str(args.warmup)]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", TK_FORCE_PROCESS_GROUP="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    try:
        pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
        if pr.returncode != 0 or not line:
            return dict(forced_group="failed (rc %d): %s" % (pr.returncode, pr.stderr[-300:]))
        d = json.loads(line[-1])
    except subprocess.TimeoutExpired:
        return dict(forced_group="timed out")
    r = d.get("rccl") or {}
    r.update(forced_group_ms_per_step=d["ms_per_step"], plain_ms_per_step=round(plain_ms, 3),
             overhead_ms=round(d["ms_per_step"] - plain_ms, 3),
             how="child process, same command with a forced one-rank RCCL process group "
                 "(TK_FORCE_PROCESS_GROUP=1); box-to-box and run-to-run spread of the plain step is about +-0.5 ms")
    return r


# ---------------------------------------------------------------------------------------------
def dry_launch(args):
    """Launcher check without GPUs (tests/test_data_parallel.py): the ranks rendezvous over gloo,
    all-reduce a gradient arena and rank 0 prints the JSON line with n_gpus = world."""
    from taiyaki_amd import models, parallel
    rank, local, world = parallel.init_from_env(backend="gloo")
    cores = parallel.pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    torch.manual_seed(3 + rank)
    net = models.mLstm_flipflop(size=8, stride=5)
    parallel.broadcast_parameters(net)
    arena = parallel.FlatGradArena(net, overlap_buckets=3)
    x = torch.randn(60, 2, 1)
    arena.zero()
    net(x).square().mean().backward()
    arena.allreduce_async()
    arena.finish()
    seen = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seen, torch.tensor([rank], dtype=torch.int64))
    t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(arena.flat)
    el = (time.perf_counter() - t0) / 5
    bucket_us = []
    for lo, hi in arena.slices():
        t1 = time.perf_counter()
        dist.all_reduce(arena.flat[lo:hi])
        bucket_us.append(round((time.perf_counter() - t1) * 1e6, 1))
    per_rank = _gather_per_rank(el * 1e3, torch.device("cpu"), world)
    dist.barrier()
    if rank == 0:
        print(json.dumps(dict(metric="launcher dry run (no GPU, gloo)", dry_launch=True, n_gpus=world,
                              config=dict(workload="config %d" % args.config),
                              ranks_seen=sorted(int(t.item()) for t in seen),
                              per_rank_ms=per_rank, cores_per_rank=len(cores),
                              rccl=dict(ranks=world, backend="gloo", bytes=arena.flat.numel() * 4,
                                        allreduce_us=round(el * 1e6, 1), overlap_buckets=len(arena._buckets),
                                        bucket_bytes=[(hi - lo) * 4 for lo, hi in arena.slices()],
                                        bucket_us=bucket_us))),
              flush=True)
    dist.destroy_process_group()


def _gather_per_rank(ms, dev, world):
    """Every rank's own ms/step: min / max / all -- a straggler (a rank whose launch thread was
    starved, a GPU that clocks lower) shows here before it shows in the scaling curve."""
    mine = torch.tensor([ms], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    vals = [round(float(t.item()), 3) for t in out]
    return dict(min=min(vals), max=max(vals), all=vals)


def variable_length_bench(args, cfg, trainer, dev, rank, use_graph, lo, hi, steps, warmup):
    """The reference's per-iteration chunk length draw (bin/train_flipflop.py:554-563) on graph replay: `--chunk-len-range
    MIN MAX` prints the record as its own line; the default N = 1 run carries a short leg of it as `varlen`."""
    from taiyaki_amd import _lib, train
    stride, cat_mod = cfg["stride"], cfg["model"] == "mLstm_cat_mod_flipflop"
    lens = sorted({train.bucket_chunk_len(x, stride, args.len_bucket) for x in range(lo, hi + 1, stride)})
    # the reference keeps samples per sub-batch constant: min_sub_batch_size * chunk_len_max / chunk_len
    nb_of = lambda cl: max(1, int(cfg["batch"] * cfg["chunk_len"] / cl + 0.5))      # noqa: E731
    by_len = {cl: make_batches(nb_of(cl), cl, stride, 17 + rank + cl, dev, n=2, spb=cfg["spb"], cat_mod=cat_mod)
              for cl in lens}
    maxlen = {cl: max(b["seqlens"].tk_max_seqlen for b in bs) for cl, bs in by_len.items()}
    stepper = trainer
    mode = "eager"
    if use_graph:
        stepper = train.GraphCacheTrainer(trainer, seq_capacity_per_chunk=lambda cl: cl // stride + 1,
                                          max_seqlen_of=lambda cl: maxlen[cl])
        mode = "one captured forward+loss graph per shape (train.GraphCacheTrainer), eager backward, AdamW replayed"
    rng = np.random.RandomState(11)
    draw = lambda: train.bucket_chunk_len(int(rng.randint(lo, hi + 1)), stride, args.len_bucket)    # noqa: E731
    t_cap = time.perf_counter()
    for cl in lens:                         # every shape once: captures
        stepper.step(by_len[cl][0])
    torch.cuda.synchronize()
    t_cap = time.perf_counter() - t_cap
    for i in range(warmup):
        stepper.step(by_len[draw()][i % 2])
    torch.cuda.synchronize()
    hits0, miss0 = getattr(stepper, "hits", 0), getattr(stepper, "misses", 0)
    chunks = samples = 0
    t0 = time.perf_counter()
    for i in range(steps):
        cl = draw()
        b = by_len[cl][i % 2]
        stepper.step(b)
        chunks += b["indata"].shape[1]
        samples += b["indata"].shape[0] * b["indata"].shape[1]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    _lib.raise_if_nonfinite()
    hits, misses = getattr(stepper, "hits", 0) - hits0, getattr(stepper, "misses", 0) - miss0
    return dict(
        metric="signal-chunks/sec, flip-flop train step with the reference's per-iteration chunk length draw",
        value=round(chunks / el, 2), unit="chunks/s", samples_per_s=round(samples / el, 1), n_gpus=1,
        steps=steps, warmup=warmup, ms_per_step=round(el / steps * 1e3, 3), higher_is_better=True,
        dtype="f32", data="synthetic",
        graph_cache=dict(distinct_graphs=len(getattr(stepper, "entries", {})), hits=hits, misses=misses,
                         hit_rate=round(hits / max(1, hits + misses), 3), capture_s=round(t_cap, 2),
                         note="every shape of the grid is captured once before the warm-up (capture_s, outside the timed "
                              "region); hits / misses count the TIMED steps"),
        config=dict(workload=cfg["label"] + ", chunk_len drawn in [%d, %d] per step, batch = %d * %d / chunk_len "
                    "(bin/train_flipflop.py:554-563)" % (lo, hi, cfg["batch"], cfg["chunk_len"]),
                    chunk_len_grid=lens, batch_of_len={str(cl): nb_of(cl) for cl in lens}, launch=mode))


def kernel_records(out, args, cfg, dev, world, T, nbatch, chunk_len, cat_mod, S):
    """The loss-path kernel records of the JSON line (`roofline*`, `loss_path`, `cpu_baseline`)."""
    # ---- loss-path kernels, HIP events on the launching stream, right after the timed steps
    #      (the step itself may be a hipGraph replay, so per-launch events cannot be
    #      interleaved with it) -----------------------------------------------------------
    step_ops = LossOps(T, nbatch, dev, realistic_chunk_len=chunk_len, spb=cfg["spb"], cat_mod=cat_mod)
    specs = [("logz", T, nbatch, 0), ("crf", T, nbatch, chunk_len)]
    rowk = None
    if not args.no_rowk:
        rowk = LossOps(4000, 256, dev)
        specs = [("logz", 4000, 256, 0), ("crf", 4000, 256, 0)] + specs
    # N > 1: the other ranks wait in a barrier while rank 0 fills in the kernel records -- no
    # rocprofv3 passes and no CPU leg there (both belong to the N = 1 line; `traffic` then
    # comes from the committed profile of the same kernel hash)
    no_pmc = args.no_pmc or world > 1
    no_cpu = args.no_cpu_baseline or world > 1
    measured = {} if no_pmc else measure_traffic_now(specs)
    khash = kernel_hash()

    def traffic_of(op, t, n, realistic):
        key = "%s:%d:%d:%d" % (op, t, n, realistic)
        if key in measured:
            return measured[key], dict(measured="in this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                                "separate passes, tools/pmc_traffic.py)", kernel_hash=khash,
                                       current=True)
        return committed_traffic(op, t, n)

    def logz_roofline(ops, reps, label):
        mean_s, min_s = _events_mean_min(ops.logz_op, reps)
        tr, src = traffic_of("logz", ops.T, ops.N, 0)
        rec = roofline_record("logZ forward-backward op (logz_transfer + logz_middle + logz_posterior), "
                              "T=%d N=%d (%s)" % (ops.T, ops.N, label), 3.0 * ops.T * ops.N * 40 * 4,
                              mean_s, min_s, reps, tr, src)
        if ops.x40.numel() * 4 >= (64 << 20):
            # the fraction against a MEASURED device-copy ceiling too -- the score tensor copied into
            # the gradient tensor by the runtime's own copy kernel, read + write bytes over the HIP-event time
            from taiyaki_amd import _lib
            Lc = _lib.lib()

            def own_copy():
                _lib.check(Lc.tk_devcopy_f32_dev(_lib.ptr(ops.lgrad), _lib.ptr(ops.x40), ops.x40.numel(), _lib.stream_ptr()),
                           "tk_devcopy_f32_dev")
            cp_mean, cp_min = _events_mean_min(own_copy, 20, warm=3)
            rt_mean, _ = _events_mean_min(lambda: ops.lgrad.copy_(ops.x40), 20, warm=3)
            ceiling = 2.0 * ops.x40.numel() * 4 / cp_mean / 1e9
            rec["copy_ceiling"] = dict(value=round(ceiling, 1), unit="GB/s", mean_us=round(cp_mean * 1e6, 2),
                                       min_us=round(cp_min * 1e6, 2),
                                       runtime_copy_GBs=round(2.0 * ops.x40.numel() * 4 / rt_mean / 1e9, 1),
                                       how="tk_devcopy_f32_dev (this library's float4 streaming copy, nontemporal loads and "
                                           "stores, 4 x 16 B in flight per lane) of the (T, N, 40) fp32 score tensor into the "
                                           "gradient tensor: (read + write bytes) / HIP-event time, 20 launches; "
                                           "runtime_copy_GBs = torch's copy_ of the same tensors")
            rec["frac_of_copy_ceiling"] = round(rec["achieved"] / ceiling, 4)
        return rec

    def crf_roofline(ops, reps, label, realistic):
        mean_s, min_s = _events_mean_min(ops.crf, reps, warm=5)
        tr, src = traffic_of("crf", ops.T, ops.N, realistic)
        rec = roofline_record("sequence CRF op (crf_band_sweep incl. the index build + crf_band_posterior + crf_band_tail: retry / log-domain redo of disowned reads), "
                              "T=%d N=%d S=%d, max L %d (%s)" % (ops.T, ops.N, ops.S, ops.maxlen, label),
                              3.0 * ops.T * ops.N * ops.S * 4, mean_s, min_s, reps, tr, src)
        # the bound that applies: instruction issue (the HBM fraction above is reported because SURVEY 8d asks for it)
        # (crf_band_pick_R: 15 chunk waves + the row maker; the plain CRF takes two cells per lane from 513 bases on)
        R = 1 if ops.maxlen <= (960 if ops.mod is not None else 512) else (2 if ops.maxlen <= 1920 else 4)
        W = -(-ops.maxlen // (64 * R))
        BK = 12 if (ops.mod is None or ops.maxlen > 704) else 8         # (crf_band_pick_block)
        fl = issue_floor("catmod" if ops.mod is not None else "crf", ops.T, ops.N, realistic, W, BK)
        if fl is not None:
            rec["issue_floor_us"] = fl["issue_floor_us"]
            rec["issue_floor"] = fl
            rec["frac_of_issue_floor"] = round(fl["issue_floor_us"] / (mean_s * 1e6), 3)
        rec["gated_reads"], rec["retried_reads"] = ops.gate_counts()
        return rec

    if rowk is not None:
        out["roofline"] = logz_roofline(rowk, 50, "north_star kernel shape")
        out["roofline_in_step"] = logz_roofline(step_ops, 30, "the train step's own launch")
        out["roofline_crf"] = dict(
            in_step=crf_roofline(step_ops, 20, "the train step's own launch, realistic lengths", chunk_len),
            rowK=crf_roofline(rowk, 5, "north_star shape, SPEED_TEST lengths 0.45-0.55 T", 0),
            note="linear-domain band sweeps (per-cell power-of-two frames) + recomputing gradient pass: both are "
                 "bound by instruction issue -- T serial steps per read, all of a read's waves on one CU -- "
                 "not by HBM; `traffic` = scores read three times, one checkpoint column + boundary cells per "
                 "time block (12 steps; cat-mod 8) written and read, the gradient written once; achieved is the algorithmic "
                 "3*T*N*S*4 bytes over the op's duration (sweeps with the index build inside + gradient pass + the tail "
                 "launch -- per-read retry, then the log domain --, which finds nothing to do on these inputs)")
    else:
        out["roofline"] = logz_roofline(step_ops, 30, "the train step's own launch")
    # ---- Viterbi (north_star: a hand-written kernel of the path; decode.py:75-115, flipflop.py:387-518) --------
    def viterbi_roofline(ops, reps, label):
        from taiyaki_amd import _lib
        L, p = _lib.lib(), _lib.ptr
        Tn, Nn = ops.T, ops.N
        fwd = torch.empty(Tn + 1, Nn, 8, dtype=torch.float32, device=dev)
        tb = torch.empty(Tn, Nn, 8, dtype=torch.int64, device=dev)
        path = torch.empty(Tn + 1, Nn, dtype=torch.int64, device=dev)
        wsb = L.tk_flipflop_viterbi_workspace_bytes(Tn, Nn, 4)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

        def full():
            _lib.check(L.tk_flipflop_viterbi_dev(p(ops.x40), Tn, Nn, 4, p(fwd), p(tb), p(path), p(ws), wsb,
                                                 _lib.stream_ptr()), "tk_flipflop_viterbi_dev")

        def path_only():
            _lib.check(L.tk_flipflop_viterbi_dev(p(ops.x40), Tn, Nn, 4, None, None, p(path), p(ws), wsb,
                                                 _lib.stream_ptr()), "tk_flipflop_viterbi_dev")

        alg = 1.0 * Tn * Nn * 40 * 4 + (Tn + 1) * Nn * 8          # forward-only bytes + the path output
        pm, pmin = _events_mean_min(path_only, reps, warm=5)
        fm, fmin = _events_mean_min(full, reps, warm=5)
        rec = roofline_record("Viterbi, path only (what bin/basecall.py:222 keeps): viterbi forward + traceback scan, "
                              "T=%d N=%d (%s)" % (Tn, Nn, label), alg, pm, pmin, reps, None, None)
        rec["full_outputs"] = dict(mean_us=round(fm * 1e6, 2), min_us=round(fmin * 1e6, 2),
                                   bytes=alg + (Tn + 1) * Nn * 8 * 4 + Tn * Nn * 8 * 8,
                                   note="fwd (T+1, N, 8) f32 and the reference's int64 traceback (T, N, 8) written too "
                                        "(decode.py:15-39): 1.6 + 8 times the score tensor of extra writes")
        rec["bound_note"] = ("serial in T by construction (a time-parallel max-plus form would re-associate fp32 adds and "
                             "break bit-exactness): bound by one wave's issue stream per read, not by HBM; the HBM "
                             "fraction is reported because SURVEY 8d names the forward-only bytes")
        return rec

    out["roofline_viterbi"] = dict(in_step=viterbi_roofline(step_ops, 20, "the train step's shape"))
    if rowk is not None:
        out["roofline_viterbi"]["rowK"] = viterbi_roofline(rowk, 10, "north_star kernel shape")

    # ---- the whole loss path in one unit ------------------------------------------------
    # one queue: what the captured train step replays (a capturing stream never forks); two queues: what an eager
    # caller of the operator gets by default (kernel B beside kernel A's sweeps, tk_flipflop_loss_overlap)
    from taiyaki_amd import _lib
    L = _lib.lib()
    prev = L.tk_flipflop_loss_overlap(0)
    lp_mean, _ = _events_mean_min(step_ops.both, 20, warm=5)
    L.tk_flipflop_loss_overlap(1)
    lp2_mean, _ = _events_mean_min(step_ops.both, 20, warm=5)
    L.tk_flipflop_loss_overlap(prev)
    assert step_ops.finite()
    out["loss_path"] = dict(unit="chunks/s through crf grad + logZ fwd-bwd at the step's shape (T=%d, N=%d, "
                                 "S=%d, realistic lengths)" % (T, nbatch, S),
                            launch=("tk_flipflop_loss_fused_dev, cat-mod form: logZ of the canonical columns first, "
                                    "folded into the cat-mod kernel's writes; one gradient tensor" if cat_mod else
                                    "tk_flipflop_loss_fused_dev: one gradient tensor"),
                            gpu_ms=round(lp_mean * 1e3, 4), gpu_chunks_per_s=round(nbatch / lp_mean, 1),
                            form="one queue (A, then B adds in place): the form the captured train step replays",
                            eager_two_queue_ms=round(lp2_mean * 1e3, 4),
                            eager_two_queue_chunks_per_s=round(nbatch / lp2_mean, 1),
                            eager_two_queue_note="the operator's default outside a graph capture: kernel B on a second "
                                                 "hardware queue beside kernel A's sweeps, folded into A's gradient pass "
                                                 "(a captured fork / join replays 150 us slower: profiles/r4_overlap_capture_probe.txt)")
    # ---- the numbers the documents quote, under the key every consumer of this line keeps (round-5 verdict: the
    #      driver's record held `roofline` and `cpu_baseline` whole and only the NAMES of the other records) ----------
    def brief(rec):
        b = dict(frac=rec["frac"], mean_us=rec["mean_us"], min_us=rec["min_us"])
        if rec.get("traffic"):
            b["traffic_ratio"] = round(float(rec["traffic"]) / float(rec["algorithmic_bytes"]), 3)
        for k in ("frac_of_issue_floor", "issue_floor_us", "gated_reads", "retried_reads", "frac_of_copy_ceiling"):
            if k in rec:
                b[k] = rec[k]
        return b
    others = {}
    if "roofline_in_step" in out:
        others["logz_in_step"] = brief(out["roofline_in_step"])
    for k in ("in_step", "rowK"):
        if k in out.get("roofline_crf", {}):
            others["crf_" + k] = brief(out["roofline_crf"][k])
        if k in out.get("roofline_viterbi", {}):
            others["viterbi_" + k] = dict(brief(out["roofline_viterbi"][k]),
                                          full_outputs_us=out["roofline_viterbi"][k]["full_outputs"]["mean_us"])
    others["loss_path_ms"] = dict(one_queue=out["loss_path"]["gpu_ms"], eager_two_queue=out["loss_path"]["eager_two_queue_ms"])
    out["roofline"]["others"] = others
    if not no_cpu:
        cb = cpu_baseline(step_ops.host)
        cb["loss_path_gpu_ms"] = out["loss_path"]["gpu_ms"]       # (the same arrays on the GPU, beside `median_ms`)
        out["cpu_baseline"] = cb
        copies = reference_copies_ms(T, nbatch, S, dev)
        cpu_ms = nbatch / cb["value"] * 1e3
        out["loss_path"].update(
            cpu_chunks_per_s=cb["value"], cpu_ms=round(cpu_ms, 3), copies_ms=round(copies, 3),
            cpu_with_copies_chunks_per_s=round(nbatch / ((cpu_ms + copies) * 1e-3), 2),
            same_inputs=True,
            note="cpu = %s on %d host threads, same arrays as gpu_ms; copies = score tensor D->H + gradient H->D (pinned), what "
                 "the reference's CPU extension adds per call (ctc.pyx:119, 139-141)" % (cb["kind"], cb["cores"]))


def rccl_debug_summary(limit=14):
    """What RCCL said about itself (NCCL_DEBUG=INFO, subsystems INIT + TUNING, one file per process): the
    distinct lines that name the algorithm / protocol / channels it chose for this job's collectives."""
    import glob
    import re
    pat = os.environ.get("NCCL_DEBUG_FILE")
    if not pat:
        return None
    mine = pat.replace("%p", str(os.getpid())).replace("%h", socket.gethostname())
    files = [mine] if os.path.exists(mine) else sorted(glob.glob(pat.replace("%p", "*").replace("%h", "*")))[:1]
    seen, out = set(), []
    for fn in files:
        try:
            for ln in open(fn, errors="replace"):
                if not re.search(r"Algo|proto|[Cc]hannel|Ring|Tree|nranks|comm .* rank", ln):
                    continue
                key = re.sub(r"^.*?NCCL INFO\s*", "", ln.strip())
                key = re.sub(r"0x[0-9a-f]+|\b\d+\.\d+\b", "#", key)        # (pointers and times differ line to line)
                if key not in seen and len(out) < limit:
                    seen.add(key)
                    out.append(re.sub(r"^.*?NCCL INFO\s*", "", ln.strip())[:200])
        except OSError:
            pass
    return dict(source="NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=%s, rank 0's file" % os.environ.get("NCCL_DEBUG_SUBSYS"),
                lines=out) if out else None


def direct_leg(args, arena, comm, rank, world, dev, timed_steps, event_time_us, nbatch, on_timeout=None, budget_s=150.0):
    """`rccl.direct` of the default N > 1 line: the C-ABI collective (own rendezvous) on the same steps.  A
    watchdog thread lets the line go out without this leg if it hangs: every rank then leaves through
    os._exit after rank 0 has printed what it has."""
    import threading
    from taiyaki_amd import parallel
    state = dict(done=False)

    def bail():
        if state["done"]:
            return
        print("[rank %d] rccl.direct leg did not finish in %.0f s: dropped" % (rank, budget_s), file=sys.stderr, flush=True)
        if on_timeout is not None:
            on_timeout()            # rank 0: the measured line, without this leg
        os._exit(0)

    timer = threading.Timer(budget_s, bail)
    timer.daemon = True
    timer.start()
    try:
        coll = parallel.DirectRccl(rank, world, exchange=lambda b: parallel.socket_rendezvous(
            rank, world, b, nbytes=len(b) if b is not None else 128), device=dev)
        # (a) the two transports on the SAME seeded gradients, bit for bit: one ProcessGroupNCCL all-reduce against one
        #     tk_allreduce_f32_dev of a copy (RCCL's ring order is a function of the communicator: two communicators over
        #     the same devices may sum in another order -- `max_abs_diff` then says how far apart, `equal` whether at all)
        equal = None
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                gen = torch.Generator(device=dev).manual_seed(4242 + rank)
                seeded = torch.randn(arena.flat.numel(), generator=gen, device=dev, dtype=torch.float32)
                via_pg, via_abi = seeded.clone(), seeded.clone()
                dist.all_reduce(via_pg, op=dist.ReduceOp.SUM)
                coll.all_reduce(via_abi).wait()
                torch.cuda.synchronize()
                same = float(torch.equal(via_pg, via_abi))
                diff = float((via_pg - via_abi).abs().max())
                # ... and every rank holds the same sum (first / last rank's checksum against this one's)
                csum = float(via_abi.double().sum())
                sums = comm.gather(csum)
                equal = dict(equal=bool(min(comm.gather(same)) == 1.0), max_abs_diff=max(comm.gather(diff)),
                             ranks_agree=bool(max(sums) == min(sums)), elements=int(seeded.numel()))
                del seeded, via_pg, via_abi
        except Exception as exc:      # noqa: BLE001 -- report, never hide
            equal = dict(failed="%s: %s" % (type(exc).__name__, str(exc)[:200]))
        arena.collective, saved_world = coll, arena.world
        try:
            us = event_time_us(lambda: arena._all_reduce(arena.flat).wait(), 20)
            el = timed_steps(max(2, args.warmup // 2), args.steps, first=7)
            el = max(comm.gather(el))
        finally:
            arena.collective, arena.world = None, saved_world
        out = dict(collective="tk_allreduce_f32_dev (C ABI over RCCL), communicator from tk_rendezvous_bytes at port %d"
                              % parallel.rendezvous_port(),
                   allreduce_us=round(float(np.mean(us)), 1), allreduce_min_us=round(us[0], 1),
                   ms_per_step=round(el / args.steps * 1e3, 3), value=round(nbatch * world * args.steps / el, 2),
                   steps=args.steps)
        if equal is not None:
            out["vs_process_group"] = equal
        torch.cuda.synchronize()
        coll.close()
        return out
    except Exception as exc:          # report, never hide -- and never lose the measured line over it
        return dict(failed="%s: %s" % (type(exc).__name__, str(exc)[:300]))
    finally:
        state["done"] = True
        timer.cancel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration, SURVEY.md section 8 numbering (default 2 = configs[1])")
    ap.add_argument("--chunk-len", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="chunks per GPU")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--conv", choices=["gemm", "miopen"], default="gemm",
                    help="evaluate the Convolution layers as unfold+GEMM (default) or nn.Conv1d")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark")
    ap.add_argument("--lstm", choices=["miopen", "native"], default="miopen",
                    help="native = ATen per-timestep LSTM (torch.backends.cudnn.enabled=False), "
                         "capturable into a hipGraph; miopen = fused MIOpen RNN")
    ap.add_argument("--graph", action="store_true",
                    help="replay the WHOLE step from a captured hipGraph (probed in a child process "
                         "first).  Default for --lstm native; with the MIOpen LSTM the RNN backward "
                         "is not capturable on ROCm 7.2 (hipBLASLt call inside capture), so the "
                         "default there is the hybrid scheme (--hybrid)")
    ap.add_argument("--no-graph", action="store_true", help="force eager launch")
    ap.add_argument("--hybrid", action="store_true",
                    help="(default with the MIOpen LSTM) replay forward + loss and AdamW from "
                         "hipGraphs, run the uncapturable RNN backward eagerly; probed in a child "
                         "process first, eager launch if the probe fails; --no-graph disables")
    ap.add_argument("--data", choices=["arena", "store"], default="arena",
                    help="arena: a few pre-assembled synthetic batches cycled from HBM (default); "
                         "store: every step samples, filters and assembles its batch on the device "
                         "from a synthetic mapped-signal set resident in HBM "
                         "(taiyaki_amd.mapped_signal, the reference's prepare_random_batches)")
    ap.add_argument("--mapped-signal", default=None, metavar="FILE",
                    help="with --data store: take the reads from this mapped-signal HDF5 (classic layout, read by "
                         "taiyaki_amd.hdf5_lite) or packed .npz file instead of synthetic reads, e.g. "
                         "tests/golden/mapped_signal/mapped_reads_0.hdf5 (real r9.4.1 reads)")
    ap.add_argument("--overlap-buckets", type=int, default=0,
                    help="gradient all-reduce slices issued from backward hooks (N > 1); 0 (default) = ONE flat "
                         "all-reduce after backward -- measured: slices issued inside the eager RNN backward cost "
                         "3-4 ms each (profiles/r4_forced_group_bisect.txt), the one flat call nothing")
    ap.add_argument("--collective", choices=["pg", "direct"], default="pg",
                    help="N > 1: what reduces the gradients.  pg (default): torch.distributed's ProcessGroupNCCL (= RCCL); "
                         "the line then also carries `rccl.direct`, the same steps re-timed with this repo's C-ABI "
                         "collective.  direct: libtaiyaki_amd_rccl.so (tk_allreduce_f32_dev) with its OWN socket "
                         "rendezvous -- no torch.distributed process group exists in the ranks at all (barriers and "
                         "the max over ranks go through the same all-reduce)")
    ap.add_argument("--no-varlen", action="store_true",
                    help="N = 1: skip the short leg with the reference's per-iteration chunk length draw (`varlen`)")
    ap.add_argument("--no-forced-group", action="store_true",
                    help="N = 1: skip the child run that repeats the step with a one-rank RCCL process group "
                         "(the `rccl.overhead_ms` field)")
    ap.add_argument("--chunk-len-range", type=int, nargs=2, default=None, metavar=("MIN", "MAX"),
                    help="the reference's own schedule (bin/train_flipflop.py:554-563, defaults 3000 8000): every "
                         "step draws a chunk length in [MIN, MAX] and rescales the batch to batch * chunk_len / "
                         "length; lengths are rounded down to a grid (--len-bucket) and every shape replays its "
                         "own captured graph (train.GraphCacheTrainer).  Prints its own JSON line")
    ap.add_argument("--len-bucket", type=int, default=100, help="grid of --chunk-len-range in blocks (strides)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rowk", action="store_true")
    ap.add_argument("--no-kernel-records", action="store_true",
                    help="the train-step line only (no roofline / loss_path / cpu_baseline records)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC passes")
    ap.add_argument("--probe-graph", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-launch", action="store_true", help=argparse.SUPPRESS)
    argv = sys.argv[1:]
    args = ap.parse_args(argv)
    cfg = dict(CONFIGS[args.config])
    if args.gpus is None:
        args.gpus = 8 if args.config == 3 and "WORLD_SIZE" not in os.environ else int(os.environ.get("WORLD_SIZE", "1"))
        if args.config == 3 and "--gpus" not in argv:
            argv = argv + ["--gpus", str(args.gpus)]
    for k, a in (("chunk_len", args.chunk_len), ("batch", args.batch), ("size", args.size)):
        if a is not None:
            cfg[k] = a

    if os.environ.get("TK_BENCH_STACKS_AFTER"):
        # debugging aid: dump every thread's Python stack to stderr after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["TK_BENCH_STACKS_AFTER"]), repeat=False)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.probe_graph:
        self_launch(args, argv)
    if args.dry_launch:
        return dry_launch(args)

    from taiyaki_amd import _lib, models, parallel, train
    if args.probe_graph:
        rank, local, world = 0, int(os.environ.get("LOCAL_RANK", "0")), 1
        if os.environ.get("TK_BENCH_SHARE_GPU"):
            local = 0
    else:
        # TK_BENCH_SHARE_GPU=1 (tests, 1-GPU boxes): every rank drives cuda:0 and the gradients are
        # reduced over gloo -- RCCL refuses two ranks on one device.  Exercises the whole N > 1
        # choreography (sharding, hooks, barrier bracket, max over ranks, rank-0 report) on the real
        # kernels; the number it prints is not a scaling measurement
        share = bool(os.environ.get("TK_BENCH_SHARE_GPU"))
        direct_only = args.collective == "direct" and not share
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not share:
            # RCCL says which algorithm / protocol / channel count it picked (ring vs tree, LL vs Simple): into a file
            # per process, summarised in the line's `rccl.debug` (must be set before any communicator exists)
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING")
            os.environ.setdefault("NCCL_DEBUG_FILE", os.path.join(
                os.environ.get("TMPDIR", "/tmp"), "tk_rccl_debug_%s.%%p.log" % os.environ.get("MASTER_PORT", "0")))
        rank, local, world = parallel.init_from_env(backend="gloo" if share else None, process_group=not direct_only)
        if share:
            local = 0
    if world != args.gpus and not args.probe_graph:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the flip-flop operators have no CPU fallback)")
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cores = parallel.pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if world > 1 else []
    _lib.lib()
    _lib.set_strict(False)      # status words are checked once, after the timed region

    stride, chunk_len, nbatch, size = cfg["stride"], cfg["chunk_len"], cfg["batch"], cfg["size"]
    cat_mod = cfg["model"] == "mLstm_cat_mod_flipflop"
    T = chunk_len // stride
    torch.manual_seed(1234)     # same init on every rank, then broadcast anyway
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    if args.lstm == "native":
        torch.backends.cudnn.enabled = False
    try:
        torch.backends.cuda.preferred_blas_library("cublas")        # = rocBLAS on ROCm
    except Exception:
        pass
    hybrid = args.lstm == "miopen" and not args.graph
    use_graph = (args.graph or args.lstm == "native" or hybrid) and not args.no_graph
    if use_graph and not args.probe_graph:
        # a failed capture aborts the process inside the HIP runtime, so try it in a child first,
        # at the real shapes (what is probed is this very capture, workspace sizes included)
        cmd = [sys.executable, os.path.abspath(__file__), "--probe-graph", "--config", str(args.config),
               "--chunk-len", str(chunk_len), "--batch", str(nbatch), "--size", str(size),
               "--conv", args.conv, "--lstm", args.lstm, "--gpus", "1"]
        cmd += ["--hybrid"] if hybrid else ["--graph"]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}
        try:
            # (well inside the process group's 10-minute collective timeout: the other ranks wait for
            # this rank's verdict in an all-reduce)
            pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
            use_graph = pr.returncode == 0 and "graph-probe-ok" in pr.stdout
            if not use_graph:
                print("[rank %d] hipGraph probe failed (rc %d): %s" % (rank, pr.returncode, pr.stderr[-400:]),
                      file=sys.stderr)
        except subprocess.TimeoutExpired:
            use_graph = False
            print("[rank %d] hipGraph probe timed out" % rank, file=sys.stderr)
    def build_net():
        if cfg["model"] == "mGru_flipflop":
            net = models.mGru_flipflop(size=size, stride=stride).to(dev)
        elif cat_mod:
            net = models.mLstm_cat_mod_flipflop(size=size, stride=stride, can_nmods=CAN_NMODS).to(dev)
        else:
            net = models.mLstm_flipflop(size=size, stride=stride).to(dev)
        for m in net.modules():
            if hasattr(m, "use_gemm"):
                m.use_gemm = args.conv == "gemm"
        return net

    net = build_net()
    # TK_RCCL_DIRECT=1: the gradient collectives through this repo's own C ABI over RCCL
    # (libtaiyaki_amd_rccl.so) instead of ProcessGroupNCCL; the process group stays up for the
    # rendezvous (it carries RCCL's unique id) and the bench's own barriers
    collective = None
    direct_only = args.collective == "direct" and not os.environ.get("TK_BENCH_SHARE_GPU") and not args.probe_graph
    if dev.type == "cuda" and ((direct_only and world > 1) or (
            os.environ.get("TK_RCCL_DIRECT") and (world > 1 or os.environ.get("TK_FORCE_PROCESS_GROUP")))):
        # (--collective direct: no process group is up, DirectRccl brings its own socket rendezvous)
        collective = parallel.DirectRccl(rank, world, device=dev, in_stream=bool(os.environ.get("TK_RCCL_INSTREAM")))
    comm = parallel.RankComm(rank, world if not args.probe_graph else 1, collective if direct_only else None, dev)
    if comm.active:
        # every rank must take the same road: the graphed and the eager trainer issue different
        # numbers of collectives while they set up (a rank whose probe failed would sit in the
        # timing barrier while the others wait for it inside the capture's warm-up all-reduce)
        agreed = min(comm.gather(1.0 if use_graph else 0.0)) > 0.5
        if use_graph and not agreed and rank == 0:
            print("hipGraph probe failed on another rank: every rank launches eagerly", file=sys.stderr)
        use_graph = agreed
    parallel.broadcast_parameters(net, collective=collective)
    arena = parallel.FlatGradArena(net, overlap_buckets=args.overlap_buckets, collective=collective)
    # the reference's default adaptive clipping (--gradient_clip_num_mads 0, window 1000):
    # gradient maxima every step, clamp once 1000 steps have been seen
    trainer = train.Trainer(net, arena, clip_num_mads=0)
    if args.chunk_len_range and not args.probe_graph:
        print(json.dumps(variable_length_bench(args, cfg, trainer, dev, rank, use_graph, args.chunk_len_range[0],
                                               args.chunk_len_range[1], args.steps, args.warmup)), flush=True)
        return
    batches = make_batches(nbatch, chunk_len, stride, 17 + rank, dev, n=2 if args.probe_graph else 4,
                           spb=cfg["spb"], cat_mod=cat_mod)
    if args.data == "store":
        for b in batches:               # (also captured: device-assembled batches may carry padding)
            b["ignore_empty"] = True
    mode = "eager"
    stepper = trainer
    if use_graph:
        try:
            cls = train.HybridGraphTrainer if hybrid else train.GraphedTrainer
            maxlen = max(b["seqlens"].tk_max_seqlen for b in batches)
            if args.data == "store":
                # device-assembled batches: the host does not see the lengths, but chunks that pass the
                # path-buffer filter (1.1 below) have fewer than chunk_len / (stride * 1.1) bases --
                # mapped_signal.sample_chunks hangs that bound on the seqlens tensor it returns
                maxlen = max(maxlen, int(chunk_len / (stride * PATH_BUFFER)) + 1)
            g = cls(trainer, batches[0], seq_capacity=nbatch * (T + 1), max_seqlen=maxlen)
            g.load(batches[0])
            g.capture()
            stepper, mode = g, ("hipGraph replay of forward+loss and of AdamW, eager backward"
                                if hybrid else "hipGraph replay of the whole step")
        except Exception as exc:      # report, never hide
            print("hipGraph capture failed (%s: %s); running eagerly" % (type(exc).__name__, exc),
                  file=sys.stderr)
    if comm.active and not args.probe_graph:
        # (same agreement after the capture itself: a rank that fell back launches eagerly everywhere)
        if min(comm.gather(1.0 if mode != "eager" else 0.0)) < 0.5 and mode != "eager":
            print("[rank %d] another rank could not capture: launching eagerly" % rank, file=sys.stderr)
            stepper, mode = trainer, "eager"
    if args.probe_graph:
        stepper.step(batches[1])
        torch.cuda.synchronize()
        _lib.raise_if_nonfinite()
        print("graph-probe-ok" if mode != "eager" else "graph-probe-eager")
        return

    next_batch = lambda i: batches[i % len(batches)]        # noqa: E731
    if args.data == "store":
        # batches straight from the file-format arrays: sample_chunks + filters + stacking +
        # flip-flop coding as three launches on this stream, nothing on the host
        from taiyaki_amd import mapped_signal, synth
        if args.mapped_signal and args.mapped_signal.endswith(".npz"):
            store = mapped_signal.MappedSignalStore.from_npz(args.mapped_signal, dev)
        elif args.mapped_signal:
            store = mapped_signal.MappedSignalStore.from_hdf5(args.mapped_signal, dev)
        else:
            store = mapped_signal.MappedSignalStore(
                synth.mapped_reads(1500, 31 + rank, mean_reflen=max(900, chunk_len // 4),
                                   long_dwell_prob=0.0003), dev)
        torch.manual_seed(99 + rank)
        fparams = store.sample_filter_parameters(1000, chunk_len, 3.0, 10.0, 0.5, stride, PATH_BUFFER)

        def next_batch(i):
            b = store.sample_chunks(nbatch, chunk_len, fparams, max_bases_per_chunk=T + 1)
            # (padding columns of a starved batch carry no sequence: keep them out of the loss)
            return dict(indata=b.indata, seqs=b.seqs, seqlens=b.seqlens, ignore_empty=True)
    def timed_steps(k_warm, k_steps, first=0):
        """W untimed steps, then EXACTLY K steps bracketed by a barrier + synchronize on both sides; returns this
        rank's seconds."""
        for i in range(k_warm):
            stepper.step(next_batch(first + i))
        comm.barrier() if comm.active else (dist.barrier() if dist.is_initialized() else None)
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        for i in range(k_steps):
            stepper.step(next_batch(first + i))
        comm.barrier() if comm.active else (dist.barrier() if dist.is_initialized() else None)
        torch.cuda.synchronize()
        return time.perf_counter() - t_start

    def event_time_us(fn, reps, warm=5):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        comm.barrier()
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) * 1e3 for a, b in evs)

    elapsed = timed_steps(args.warmup, args.steps)
    _lib.raise_if_nonfinite()
    rccl = None
    per_rank = None
    grouped = comm.active or dist.is_initialized()          # (a forced one-rank group counts: N = 1 overhead child)
    if grouped:
        mine_ms = elapsed / args.steps * 1e3
        vals = [round(v, 3) for v in comm.gather(mine_ms)] if comm.active else [round(mine_ms, 3)]
        per_rank = dict(min=min(vals), max=max(vals), all=vals)
        elapsed = max(comm.gather(elapsed)) if comm.active else elapsed
        # the gradient all-reduce on its own, event-timed on the launching stream (waiting on the work object
        # makes the current stream wait for RCCL's), whole and slice by slice as the step issues it
        reduce_flat = lambda t=arena.flat: arena._all_reduce(t).wait()       # noqa: E731
        us = event_time_us(reduce_flat, 20)
        bucket_us = []
        for lo, hi in arena.slices():
            ub = event_time_us(lambda t=arena.flat[lo:hi]: arena._all_reduce(t).wait(), 10, warm=1)
            bucket_us.append(round(float(np.mean(ub)), 1))
        rccl = dict(ranks=world,
                    backend=("none (no torch.distributed group: own socket rendezvous)" if direct_only or not dist.is_initialized()
                             else dist.get_backend()),
                    bytes=arena.flat.numel() * 4,
                    collective=("tk_allreduce_f32_dev (C ABI over RCCL, libtaiyaki_amd_rccl.so)" if arena.collective is not None
                                else "torch.distributed ProcessGroupNCCL"),
                    rendezvous=("tk_rendezvous_bytes at %s:%d (plain sockets)" % (os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                                                                 parallel.rendezvous_port())
                                if arena.collective is not None and not dist.is_initialized() else
                                "torch.distributed store at MASTER_ADDR:MASTER_PORT"),
                    allreduce_us=round(float(np.mean(us)), 1), allreduce_min_us=round(us[0], 1),
                    overlap_buckets=len(arena._buckets),
                    bucket_bytes=[(hi - lo) * 4 for lo, hi in arena.slices()], bucket_us=bucket_us,
                    note=("one flat fp32 gradient arena; in the step it is reduced in %d slices issued from backward "
                          "hooks on RCCL's high-priority stream" % len(arena._buckets) if arena._buckets else
                          "one flat fp32 gradient arena, reduced by ONE all-reduce after backward"))
        rccl["debug"] = rccl_debug_summary()
        # (b) every rank started from rank 0's weights: the parameters' checksum per rank after the timed steps (identical
        #     gradients after the all-reduce + identical optimiser steps keep them identical; any difference is a lost or
        #     reordered collective); (c) where each rank's host thread ran
        psum = float(sum(float(p.detach().double().sum()) for p in net.parameters()))
        sums = comm.gather(psum) if comm.active else [psum]
        rccl["param_checksum"] = dict(all=[float(v) for v in sums], ranks_agree=bool(max(sums) == min(sums)))
        spread = per_rank["max"] - per_rank["min"]
        rccl["per_rank_spread_ms"] = round(spread, 3)
        try:
            rccl["host_cores_rank0"] = sorted(os.sched_getaffinity(0))[:64]
        except (AttributeError, OSError):
            rccl["host_cores_rank0"] = None

    if rank == 0:
        nglobal = nbatch * world
        S = 46 if cat_mod else 40
        out = dict(metric="signal-chunks/sec (T=4000) flip-flop train step", value=round(
                       nglobal * args.steps / elapsed, 2),
                   unit="chunks/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32",
                   data=("reads of " + os.path.basename(args.mapped_signal)
                         if args.data == "store" and args.mapped_signal else "synthetic"),
                   config=dict(workload="%s, chunk_len=%d (T=%d blocks), %d chunks/GPU, size %d, HIP flip-flop %s "
                               "loss + logZ, gradient maxima/clipping, AdamW"
                               % (cfg["label"], chunk_len, T, nbatch, size, "cat-mod" if cat_mod else "CRF"),
                               config_id=args.config, model=cfg["model"],
                               global_batch=nglobal, chunk_len=chunk_len, launch=mode,
                               hw_queues=int(os.environ.get("GPU_MAX_HW_QUEUES", "0")),
                               conv=args.conv, lstm=args.lstm,
                               batches=("assembled on the device every step from a mapped-signal set in HBM"
                                        if args.data == "store" else "pre-assembled, cycled from HBM"),
                               parallelism="dp%d (reads sharded; %s)" % (
                                   world, "gradient all-reduce in %d slices from backward hooks" % args.overlap_buckets
                                   if args.overlap_buckets > 1 else "one flat RCCL gradient all-reduce after backward")))
        if rccl is not None:
            out["rccl"] = rccl
            out["per_rank_ms"] = per_rank
            out["cores_per_rank"] = len(cores)
        elif world == 1 and not args.no_forced_group and not dist.is_initialized():
            out["rccl"] = forced_group_overhead(args, argv, elapsed / args.steps * 1e3)
        if not args.no_kernel_records:
            kernel_records(out, args, cfg, dev, world, T, nbatch, chunk_len, cat_mod, S)
            if world == 1 and not args.no_varlen and not grouped:
                # the schedule the reference actually trains with (chunk_len drawn per iteration, defaults 3000 .. 8000):
                # ten driver-timed steps on the per-shape graph cache (round-4 verdict item 7)
                try:
                    net2 = build_net()          # (a trainer of its own: the captured step above keeps its optimiser state)
                    trainer2 = train.Trainer(net2, parallel.FlatGradArena(net2), clip_num_mads=0)
                    out["varlen"] = variable_length_bench(args, cfg, trainer2, dev, rank, mode != "eager", 3000, 8000, 10, 3)
                except Exception as exc:          # report, never hide
                    out["varlen"] = dict(failed="%s: %s" % (type(exc).__name__, str(exc)[:300]))
    else:
        out = None
    def emit(line):
        # RCCL writes a version banner through C stdio (fully buffered when stdout is a pipe):
        # flush it first so that the JSON line is the LAST thing rank 0 prints
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)

    comm.barrier()
    if (comm.active and world > 1 and arena.collective is None and dev.type == "cuda"
            and not os.environ.get("TK_BENCH_SHARE_GPU") and not os.environ.get("TK_BENCH_NO_DIRECT_LEG")):
        # Both transports in one go: the SAME steps again with the C-ABI collective behind its own socket
        # rendezvous (a second communicator next to the process group's).  Guarded: this leg has never seen
        # more than one GPU where it was written -- if it does not finish in time the line goes out without it.
        res = direct_leg(args, arena, comm, rank, world, dev, timed_steps, event_time_us, nbatch,
                         on_timeout=(lambda: emit(dict(out, rccl=dict(out["rccl"], direct=dict(failed="timed out")))))
                         if out is not None else None)
        if out is not None:
            out["rccl"]["direct"] = res
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        emit(out)


if __name__ == "__main__":
    main()
