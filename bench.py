#!/usr/bin/env python
"""bench.py -- flip-flop train-step throughput + loss-kernel roofline on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one optimiser step of the reference's flip-flop trainer
(bin/train_flipflop.py:544-622) on BASELINE.json configs[1]: mLstm_flipflop
(size 256, stride 5, winlen 19), chunk_len 4000 (T = 800 blocks), 128 chunks per
GPU: Conv/LSTM stack in PyTorch-ROCm fp32 -> HIP flip-flop CRF loss + HIP logZ
-> backward -> ONE flat RCCL all-reduce -> AdamW.  Synthetic chunks (resident in
HBM before the timed region), random-init weights.  Weak scaling: per-GPU batch
is fixed, value = all ranks' chunks / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":     the logZ forward-backward op (K1+K2+K3 launches) timed with HIP
                  events on its own stream inside the timed train steps;
                  achieved = 3*T*N*S*4 bytes / mean duration
  "roofline_rowK": the same op at the north_star kernel shape T=4000 / N=256
  "cpu_baseline": the reference C (oracle/_ref) or the oracle port on host cores.
"""
import argparse
import json
import os
import sys
import time

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
                        seqs=torch.from_numpy(seqs), seqlens=torch.from_numpy(seqlens)))
    return out


def time_logz_op(T, N, dev, reps, seed=1):
    """Mean duration (s) of the logZ forward-backward op at (T, N) via HIP events."""
    from taiyaki_amd import layers, synth
    x = torch.from_numpy(synth.scores(T, N, 40, seed)).to(dev)
    for _ in range(3):
        layers._logz_launch(x, True)
    torch.cuda.synchronize()
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
    threads = min(cores, 8)     # the reference's own advice: OMP_NUM_THREADS=8 (README.md:362-372)
    oracle.set_threads(threads)
    inp = synth.crf_case(T, N, 1)
    use_ref = oracle.ref_available()
    oracle.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0, use_ref=use_ref)
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.crf_flipflop_loss(inp["scores"], inp["seqs"], inp["seqlens"], 1.0, use_ref=use_ref)
        oracle.flipflop_logz_grad(inp["scores"])
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 50:
            break
    return dict(value=round(N * reps / el, 2), unit="chunks/s (loss path only: crf grad + logZ fwd-bwd)",
                cores=threads, kind="reference" if use_ref else "port",
                sample="%d reps of T=%d N=%d (cfg 2 shape, SPEED_TEST inputs), %.1f s; host has %d cores; "
                       "A = %s, B = oracle port" % (reps, T, N, el, cores,
                                                    "genuine reference C (oracle/_ref)" if use_ref
                                                    else "oracle port"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chunk-len", type=int, default=4000)
    ap.add_argument("--batch", type=int, default=128, help="chunks per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rowk", action="store_true")
    args = ap.parse_args()

    from taiyaki_amd import _lib, layers, models, parallel, train
    rank, local, world = parallel.init_from_env()
    if world != args.gpus:
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
    net = models.mLstm_flipflop(size=args.size, stride=stride).to(dev)
    parallel.broadcast_parameters(net)
    arena = parallel.FlatGradArena(net)
    trainer = train.Trainer(net, arena)
    batches = make_batches(args.batch, args.chunk_len, stride, 17 + rank, dev)

    # ---- HIP-event instrumentation of the logZ op inside the train step ------
    events = []
    orig_launch = layers._logz_launch
    timing = {"on": False}

    def timed_launch(x, want_grad):
        if not timing["on"]:
            return orig_launch(x, want_grad)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()      # torch's current stream == the stream the kernels are launched on
        r = orig_launch(x, want_grad)
        b.record()
        events.append((a, b))
        return r
    layers._logz_launch = timed_launch

    for i in range(args.warmup):
        trainer.step(batches[i % len(batches)])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timing["on"] = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.step(batches[i % len(batches)])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing["on"] = False
    layers._logz_launch = orig_launch
    _lib.raise_if_nonfinite()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        nglobal = args.batch * world
        ms = [a.elapsed_time(b) for a, b in events]
        dur = float(np.mean(ms)) * 1e-3
        alg = 3.0 * T * args.batch * 40 * 4
        roofline = dict(bound="hbm", kernel="logZ forward-backward op (logz_transfer + logz_scan + "
                        "logz_posterior), T=%d N=%d in-step" % (T, args.batch),
                        achieved=round(alg / dur / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(alg / dur / 1e9 / HBM_PEAK_GBS, 4), traffic=None,
                        algorithmic_bytes=alg, mean_us=round(dur * 1e6, 2), launches=len(ms))
        out = dict(metric="signal-chunks/sec (T=4000) flip-flop train step", value=round(
                       nglobal * args.steps / elapsed, 2),
                   unit="chunks/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="configs[1]: mLstm_flipflop r9.4.1 DNA, chunk_len=%d (T=%d "
                               "blocks), %d chunks/GPU, size %d, HIP flip-flop CRF loss + logZ, AdamW"
                               % (args.chunk_len, T, args.batch, args.size),
                               global_batch=nglobal, chunk_len=args.chunk_len,
                               parallelism="dp%d (reads sharded, flat RCCL all-reduce)" % world),
                   roofline=roofline)
        if not args.no_rowk:
            mean_s, min_s = time_logz_op(4000, 256, dev, 20)
            algk = 3.0 * 4000 * 256 * 40 * 4
            out["roofline_rowK"] = dict(
                bound="hbm", kernel="logZ forward-backward op, T=4000 N=256 (north_star kernel shape)",
                achieved=round(algk / mean_s / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(algk / mean_s / 1e9 / HBM_PEAK_GBS, 4), traffic=None,
                algorithmic_bytes=algk, mean_us=round(mean_s * 1e6, 2), min_us=round(min_s * 1e6, 2))
            out["crf_op_ms"] = dict(cfg2=round(time_crf_op(T, args.batch, dev, 5) * 1e3, 3))
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, args.batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
