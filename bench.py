#!/usr/bin/env python
"""bench.py -- flip-flop train-step throughput + loss-kernel roofline on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one optimiser step of the reference's flip-flop trainer
(bin/train_flipflop.py:544-622) on BASELINE.json configs[1]: mLstm_flipflop
(size 256, stride 5, winlen 19), chunk_len 4000 (T = 800 blocks), 128 chunks per
GPU: Conv/LSTM stack in PyTorch-ROCm fp32 -> HIP flip-flop CRF loss + HIP logZ
-> backward -> ONE flat RCCL all-reduce -> AdamW.  Forward + loss and AdamW are
replayed from hipGraphs, the MIOpen RNN backward (not capturable) is launched eagerly.  Synthetic chunks (resident in
HBM before the timed region), random-init weights.  Weak scaling: per-GPU batch
is fixed, value = all ranks' chunks / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     the logZ forward-backward op (3 launches) on the tensor BASELINE.json's
                  north_star names for the roofline target (T=4000 blocks, N=256 reads),
                  timed with HIP events on its launching stream;
                  achieved = 3*T*N*S*4 bytes / mean duration; traffic = HBM
                  bytes of the committed rocprofv3 PMC passes (profiles/)
  "roofline_in_step": the same op at the shape the train step itself launches (T=800, N=128)
  "cpu_baseline": the reference C (oracle/_ref) or the oracle port on host cores.
"""
import argparse
import json
import os
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

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)


def make_batches(nbatch, chunk_len, stride, seed, dev, n=4):
    from taiyaki_amd import synth
    T = chunk_len // stride
    out = []
    for i in range(n):
        s = seed * 1000 + i
        seqlens = synth.realistic_seqlens(T, nbatch, s, chunk_len, 9.0)
        seqs, _ = synth.sequences(seqlens, s)
        sig = synth.signal_chunks(chunk_len, nbatch, s)
        out.append(dict(indata=torch.from_numpy(sig).to(dev),
                        seqs=torch.from_numpy(seqs).to(device=dev, dtype=torch.int32),
                        seqlens=torch.from_numpy(seqlens).to(device=dev, dtype=torch.int32)))
    return out


def pmc_traffic(T, N):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in
    separate runs, gfx950 corrections applied -- profiles/r1_pmc_logz_*_traffic.json); None when
    no counters were collected for this shape."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in sorted(os.listdir(os.path.join(here, "profiles"))) if os.path.isdir(os.path.join(here, "profiles")) else []:
        if name.startswith("r1_pmc_logz_") and name.endswith("_traffic.json"):
            with open(os.path.join(here, "profiles", name)) as fh:
                d = json.load(fh)
            if d["shape"]["T"] == T and d["shape"]["N"] == N:
                return d["traffic_bytes"]
    return None


def time_logz_op(T, N, dev, reps, seed=1):
    """Mean duration (s) of the logZ forward-backward op at (T, N) via HIP events."""
    from taiyaki_amd import layers, synth
    x = torch.from_numpy(synth.scores(T, N, 40, seed)).to(dev)
    for _ in range(20):         # steady state: clocks and caches settle over the first ~15 launches
        layers._logz_launch(x, True)
    torch.cuda.synchronize()
    # Keep the GPU busy while the host enqueues the timed launches: the ~40 us of Python between
    # an event record and the first kernel launch must not show up as GPU idle time inside the
    # event window (the events then bracket exactly the three kernels, back to back).
    torch.cuda._sleep(int(2.0e6 * max(1, reps // 10)))
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        layers._logz_launch(x, True)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return float(np.mean(ms)) * 1e-3, float(ms[0]) * 1e-3


def time_crf_op(T, N, dev, reps, seed=1):
    from taiyaki_amd import ctc, synth
    inp = synth.crf_case(T, N, seed)
    x = torch.from_numpy(inp["scores"]).to(dev)
    seqs, seqlens = torch.from_numpy(inp["seqs"]), torch.from_numpy(inp["seqlens"])
    for _ in range(2):
        ctc._run(x, seqs, seqlens, 1.0, 1.0, 1.0, 40, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctc._run(x, seqs, seqlens, 1.0, 1.0, 1.0, 40, True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def cpu_baseline(T, N, budget_s=12.0):
    """The loss path (A: crf grad, B: logZ fwd-bwd) on the host cores, same shape
    as one GPU's share of a train step.  A runs on the genuine reference C when
    oracle/_ref was built (kind = "reference"), else on the oracle port."""
    import oracle
    from taiyaki_amd import synth
    oracle.build()
    cores = os.cpu_count() or 1
    inp = synth.crf_case(T, N, 1)
    use_ref = oracle.ref_available()

    def run(threads, budget):
        oracle.set_threads(threads)
        oracle.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0, use_ref=use_ref)
        reps, t0 = 0, time.perf_counter()
        while True:
            oracle.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0, use_ref=use_ref)
            oracle.flipflop_logz_grad(inp["scores"])
            reps += 1
            el = time.perf_counter() - t0
            if el > budget:
                return reps, el

    threads = min(cores, 8)     # the reference's own advice: OMP_NUM_THREADS=8 (README.md:362-372)
    reps, el = run(threads, budget_s)
    out = dict(value=round(N * reps / el, 2),
               unit="chunks/s (loss path only: crf grad + logZ fwd-bwd)",
               cores=threads, kind="reference" if use_ref else "port",
               sample="%d reps of T=%d N=%d (cfg 2 shape, SPEED_TEST inputs), %.1f s; host has %d "
                      "cores; A = %s, B = oracle port" % (
                          reps, T, N, el, cores,
                          "genuine reference C (oracle/_ref)" if use_ref else "oracle port"))
    if cores > threads:
        allc = min(cores, N)
        reps2, el2 = run(allc, budget_s / 2)
        out["value_all_cores"] = round(N * reps2 / el2, 2)
        out["cores_all"] = allc
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chunk-len", type=int, default=4000)
    ap.add_argument("--batch", type=int, default=128, help="chunks per GPU")
    ap.add_argument("--size", type=int, default=256)
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rowk", action="store_true")
    ap.add_argument("--probe-graph", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    from taiyaki_amd import _lib, layers, models, parallel, train
    if args.probe_graph:
        rank, local, world = 0, int(os.environ.get("LOCAL_RANK", "0")), 1
    else:
        rank, local, world = parallel.init_from_env()
    if world != args.gpus and not args.probe_graph:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the flip-flop operators have no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.lib()
    _lib.set_strict(False)      # status words are checked once, after the timed region

    stride = 5
    T = args.chunk_len // stride
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
        cmd = [sys.executable, os.path.abspath(__file__), "--probe-graph", "--chunk-len",
               str(args.chunk_len), "--batch", str(args.batch), "--size", str(args.size),
               "--conv", args.conv, "--lstm", args.lstm]
        cmd += ["--hybrid"] if hybrid else ["--graph"]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}
        try:
            pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            use_graph = pr.returncode == 0 and "graph-probe-ok" in pr.stdout
            if not use_graph and rank == 0:
                print("hipGraph probe failed (rc %d): %s" % (pr.returncode, pr.stderr[-400:]),
                      file=sys.stderr)
        except subprocess.TimeoutExpired:
            use_graph = False
    net = models.mLstm_flipflop(size=args.size, stride=stride).to(dev)
    for m in net.modules():
        if hasattr(m, "use_gemm"):
            m.use_gemm = args.conv == "gemm"
    parallel.broadcast_parameters(net)
    arena = parallel.FlatGradArena(net)
    # the reference's default adaptive clipping (--gradient_clip_num_mads 0, window 1000):
    # gradient maxima every step, clamp once 1000 steps have been seen
    trainer = train.Trainer(net, arena, clip_num_mads=0)
    batches = make_batches(args.batch, args.chunk_len, stride, 17 + rank, dev,
                           n=2 if args.probe_graph else 4)
    mode = "eager"
    stepper = trainer
    if use_graph:
        try:
            cls = train.HybridGraphTrainer if hybrid else train.GraphedTrainer
            g = cls(trainer, batches[0], seq_capacity=args.batch * (T + 1))
            g.load(batches[0])
            g.capture()
            stepper, mode = g, ("hipGraph replay of forward+loss and of AdamW, eager backward"
                                if hybrid else "hipGraph replay of the whole step")
        except Exception as exc:      # report, never hide
            print("hipGraph capture failed (%s: %s); running eagerly" % (type(exc).__name__, exc),
                  file=sys.stderr)
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
        store = mapped_signal.MappedSignalStore(
            synth.mapped_reads(1500, 31 + rank, mean_reflen=max(900, args.chunk_len // 4),
                               long_dwell_prob=0.0003), dev)
        torch.manual_seed(99 + rank)
        fparams = store.sample_filter_parameters(1000, args.chunk_len, 3.0, 10.0, 0.5, stride, 1.1)

        def next_batch(i):
            b = store.sample_chunks(args.batch, args.chunk_len, fparams, max_bases_per_chunk=T + 1)
            return dict(indata=b.indata, seqs=b.seqs, seqlens=b.seqlens)
    for i in range(args.warmup):
        stepper.step(next_batch(i))
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        stepper.step(next_batch(i))
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _lib.raise_if_nonfinite()
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        nglobal = args.batch * world
        # Roofline of the loss path's HBM-bound operator (logZ forward-backward), HIP events
        # on the launching stream, right after the timed steps (the step itself may be a
        # hipGraph replay, so per-launch events cannot be interleaved with it).
        #   roofline          : the tensor BASELINE.json's north_star quotes the target on
        #                       (T=4000 blocks x N=256 reads x 40 transitions)
        #   roofline_in_step  : the very launch the train step makes (configs[1]: T=800, N=128)
        def logz_roofline(t, n, reps, label):
            mean_s, min_s = time_logz_op(t, n, dev, reps)
            alg = 3.0 * t * n * 40 * 4
            return dict(bound="hbm", kernel="logZ forward-backward op (logz_transfer + logz_middle + "
                        "logz_posterior), T=%d N=%d (%s)" % (t, n, label),
                        achieved=round(alg / mean_s / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(alg / mean_s / 1e9 / HBM_PEAK_GBS, 4), traffic=pmc_traffic(t, n),
                        algorithmic_bytes=alg, mean_us=round(mean_s * 1e6, 2),
                        min_us=round(min_s * 1e6, 2), launches=reps)
        out = dict(metric="signal-chunks/sec (T=4000) flip-flop train step", value=round(
                       nglobal * args.steps / elapsed, 2),
                   unit="chunks/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="configs[1]: mLstm_flipflop r9.4.1 DNA, chunk_len=%d (T=%d "
                               "blocks), %d chunks/GPU, size %d, HIP flip-flop CRF loss + logZ, gradient maxima/clipping, AdamW"
                               % (args.chunk_len, T, args.batch, args.size),
                               global_batch=nglobal, chunk_len=args.chunk_len, launch=mode,
                               hw_queues=int(os.environ.get("GPU_MAX_HW_QUEUES", "0")),
                               conv=args.conv, lstm=args.lstm,
                               batches=("assembled on the device every step from a mapped-signal set in HBM"
                                        if args.data == "store" else "pre-assembled, cycled from HBM"),
                               parallelism="dp%d (reads sharded, flat RCCL all-reduce)" % world))
        if not args.no_rowk:
            out["roofline"] = logz_roofline(4000, 256, 50, "north_star kernel shape")
            out["roofline_in_step"] = logz_roofline(T, args.batch, 30, "the train step's own launch")
            out["crf_op_ms"] = dict(cfg2=round(time_crf_op(T, args.batch, dev, 5) * 1e3, 3))
        else:
            out["roofline"] = logz_roofline(T, args.batch, 30, "the train step's own launch")
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, args.batch)
    else:
        out = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        # RCCL writes a version banner through C stdio (fully buffered when stdout is a pipe):
        # flush it first so that the JSON line is the LAST thing rank 0 prints
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
