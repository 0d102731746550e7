#!/usr/bin/env python
"""Plumbing experiment on the LSTM-bound train step (round-2 verdict item 8): does running the step
as TWO independent half-batch graphs on two hardware queues fill the half of the chip that the
128-workgroup per-timestep GEMMs of the MIOpen LSTM leave idle?

Same math as the reference's sub-batches (bin/train_flipflop.py:153-198): two forward + loss graphs
on 64 chunks each (separate static inputs, gradients accumulate into one arena), against one graph on
128 chunks.  Timed separately: the replayed forward + loss (GPU-bound) and the eager backward
(host-launch-bound: ~8,000 launches from one thread per backward, two backwards = twice as many).

    python tools/halfbatch_probe.py [--queues 2]        (GPU_MAX_HW_QUEUES for the compute streams)
"""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--queues", type=int, default=2)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
os.environ["GPU_MAX_HW_QUEUES"] = str(args.queues)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (BLAS environment before torch loads)
import torch  # noqa: E402
from taiyaki_amd import _lib, ctc, models, parallel, train  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _lib.set_strict(False)
    try:
        torch.backends.cuda.preferred_blas_library("cublas")
    except Exception:
        pass
    cfg = bench.CONFIGS[2]
    stride, chunk_len, size = cfg["stride"], cfg["chunk_len"], cfg["size"]
    T = chunk_len // stride
    torch.manual_seed(1234)
    net = models.mLstm_flipflop(size=size, stride=stride).to(dev)
    for m in net.modules():
        if hasattr(m, "use_gemm"):
            m.use_gemm = True
    arena = parallel.FlatGradArena(net)
    trainer = train.Trainer(net, arena, clip_num_mads=None)
    full = bench.make_batches(128, chunk_len, stride, 17, dev, n=1)[0]
    halves = bench.make_batches(64, chunk_len, stride, 18, dev, n=2)

    def capture(batch):
        g = train.GraphedTrainer(trainer, batch, seq_capacity=batch["indata"].shape[1] * (T + 1),
                                 max_seqlen=batch["seqlens"].tk_max_seqlen)
        g.load(batch)
        for _ in range(2):
            loss, _ = train.calculate_loss(net, **g.static)
            ctc.backward_unit(loss)
        torch.cuda.synchronize()
        g.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g.graph, capture_error_mode="thread_local"):
            g.loss, _ = train.calculate_loss(net, **g.static)
        torch.cuda.synchronize()
        return g

    g128 = capture(full)
    ga, gb = capture(halves[0]), capture(halves[1])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def fwd_full():
        g128.graph.replay()

    def fwd_halves_serial():
        ga.graph.replay()
        gb.graph.replay()

    def fwd_halves_two_streams():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ga.graph.replay()
        with torch.cuda.stream(s2):
            gb.graph.replay()
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    def bwd(gs):
        def fn():
            for g in gs:
                ctc.backward_unit(g.loss, retain_graph=True)
        return fn

    def timed(fn, pre=None):
        ts = []
        for _ in range(args.reps + 2):
            if pre is not None:
                pre()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return sum(ts[2:]) / args.reps

    print("GPU_MAX_HW_QUEUES=%d  (ms, mean of %d)" % (args.queues, args.reps))
    f128 = timed(fwd_full)
    fser = timed(fwd_halves_serial)
    fpar = timed(fwd_halves_two_streams)
    print("forward + loss replay:  one graph of 128 chunks %7.2f | two graphs of 64, one after the other %7.2f | "
          "two graphs of 64 on two streams %7.2f" % (f128, fser, fpar))
    b128 = timed(bwd([g128]), pre=fwd_full)
    b64 = timed(bwd([ga, gb]), pre=fwd_halves_serial)
    print("eager backward:         128 chunks %7.2f | 2 x 64 chunks %7.2f" % (b128, b64))
    print("step without optimiser: one graph %7.2f | two half graphs on two streams %7.2f" % (f128 + b128, fpar + b64))
    _lib.raise_if_nonfinite()


if __name__ == "__main__":
    main()
