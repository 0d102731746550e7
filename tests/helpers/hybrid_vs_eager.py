"""Child process of tests/test_gpu_parity.py::test_hybrid_graph_trainer_matches_eager: a failed
hipGraph capture aborts inside the HIP runtime, so the comparison runs out of process."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402  (first: it sets the BLAS environment before torch loads; make_batches)
import torch  # noqa: E402
from taiyaki_amd import _lib, models, parallel, train  # noqa: E402


def cache_main():
    """GraphCacheTrainer over the reference's variable chunk length: six steps over three lengths
    (batch size rescaled like bin/train_flipflop.py:558-563) against the eager Trainer."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _lib.set_strict(False)
    try:
        torch.backends.cuda.preferred_blas_library("cublas")
    except Exception:
        pass
    stride, size = 5, 32
    torch.manual_seed(7)
    net_a = models.mLstm_flipflop(size=size, stride=stride).to(dev)
    net_b = copy.deepcopy(net_a)
    lens = [train.bucket_chunk_len(x, stride, 20) for x in (430, 655, 810)]        # -> 400, 600, 800
    assert lens == [400, 600, 800]
    by_len = {cl: bench.make_batches(int(6 * 800 / cl + 0.5), cl, stride, 5 + cl, dev, n=2) for cl in lens}
    tr_a = train.Trainer(net_a, parallel.FlatGradArena(net_a), clip_num_mads=None)
    tr_b = train.Trainer(net_b, parallel.FlatGradArena(net_b), clip_num_mads=None)
    maxlen = {cl: max(b["seqlens"].tk_max_seqlen for b in bs) for cl, bs in by_len.items()}
    gc = train.GraphCacheTrainer(tr_b, seq_capacity_per_chunk=lambda cl: cl // stride + 1,
                                 max_seqlen_of=lambda cl: maxlen[cl])
    worst = 0.0
    order = [400, 600, 400, 800, 600, 800, 400]
    for i, cl in enumerate(order):
        b = by_len[cl][i % 2]
        la = float(tr_a.step(b))
        lb = float(gc.step(b))
        worst = max(worst, abs(la - lb) / max(1e-6, abs(la)))
    torch.cuda.synchronize()
    _lib.raise_if_nonfinite()
    pw = max(float((pa - pb).abs().max()) for pa, pb in zip(net_a.parameters(), net_b.parameters()))
    print("hybrid-ok loss_rel=%.3e param_abs=%.3e graphs=%d" % (worst, pw, len(gc.entries)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "cache":
        return cache_main()
    whole = len(sys.argv) > 1 and sys.argv[1] == "whole"
    if whole:
        # the ATen per-timestep LSTM: the only RNN whose backward captures (bench.py --lstm native)
        torch.backends.cudnn.enabled = False
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _lib.set_strict(False)
    try:
        torch.backends.cuda.preferred_blas_library("cublas")        # = rocBLAS on ROCm (as bench.py)
    except Exception:
        pass
    chunk_len, stride, nbatch, size = 400, 5, 8, 32
    T = chunk_len // stride
    torch.manual_seed(7)
    net_a = models.mLstm_flipflop(size=size, stride=stride).to(dev)
    net_b = copy.deepcopy(net_a)
    batches = bench.make_batches(nbatch, chunk_len, stride, 5, dev, n=3)
    # adaptive clipping on, with a 3-step window so that the clamp is active inside the test
    mads = None if os.environ.get("TK_TEST_NOCLIP") else 0
    tr_a = train.Trainer(net_a, parallel.FlatGradArena(net_a), clip_num_mads=mads, clip_window=3)
    tr_b = train.Trainer(net_b, parallel.FlatGradArena(net_b), clip_num_mads=mads, clip_window=3)
    cls = train.GraphedTrainer if whole else train.HybridGraphTrainer
    hy = cls(tr_b, batches[0], seq_capacity=nbatch * (T + 1))
    hy.load(batches[0])
    hy.capture(warmup=1)        # one eager step + the capture's own tail step, both on batch 0
    if whole:
        # (a whole-step capture only records: the reference trainer is one step ahead by the
        # warm-up step alone; its clipper has seen one set of maxima, like the captured one)
        la = [float(tr_a.step(batches[0]))]
    else:
        la = [float(tr_a.step(batches[0])), float(tr_a.step(batches[0]))]
    worst = 0.0
    for i in range(1, 6):
        b = batches[i % 3]
        la.append(float(tr_a.step(b)))
        lb = float(hy.step(b))
        worst = max(worst, abs(la[-1] - lb) / max(1e-6, abs(la[-1])))
    torch.cuda.synchronize()
    _lib.raise_if_nonfinite()
    if os.environ.get("TK_TEST_DEBUG") and tr_a.clipper is not None:
        print("thresh a", tr_a.clipper.thresh[:6].tolist(), "b", tr_b.clipper.thresh[:6].tolist())
        print("maxs a", tr_a.clipper.last_maxs[:6], "b", tr_b.clipper.last_maxs[:6])
        print("iters", tr_a.clipper.rolling._curr_iter, tr_b.clipper.rolling._curr_iter)
    pw = max(float((pa - pb).abs().max()) for pa, pb in zip(net_a.parameters(), net_b.parameters()))
    lr_note = ""
    if not whole:
        # the replayed AdamW reads its learning rate from a device scalar: a schedule's new value (the
        # reference steps one every iteration, bin/train_flipflop.py:605-607) must reach the NEXT replay --
        # the same step taken by the eager trainer with its param group's lr set the usual way
        for g in tr_a.opt.param_groups:
            g["lr"] = 1e-3
        hy.set_lr(1e-3)
        la2, lb2 = float(tr_a.step(batches[1])), float(hy.step(batches[1]))
        torch.cuda.synchronize()
        pw2 = max(float((pa - pb).abs().max()) for pa, pb in zip(net_a.parameters(), net_b.parameters()))
        # ... and a rate of 0 freezes the weights (weight decay is lr * wd)
        before = [p.detach().clone() for p in net_b.parameters()]
        hy.set_lr(0.0)
        hy.step(batches[2])
        torch.cuda.synchronize()
        frozen = max(float((p - q).abs().max()) for p, q in zip(net_b.parameters(), before))
        lr_note = " lr_step_param_abs=%.3e lr0_moved=%.3e" % (pw2, frozen)
        assert abs(la2 - lb2) <= 1e-4 * abs(la2) and pw2 < 1e-4 and frozen == 0.0, lr_note
    if whole and mads is not None:
        assert tr_b.clipper.active, "the captured step never received clipping thresholds"
    print("hybrid-ok loss_rel=%.3e param_abs=%.3e%s losses=%s" % (worst, pw, lr_note, ["%.5f" % x for x in la]))


if __name__ == "__main__":
    main()
