"""Child process of tests/test_gpu_parity.py::test_hybrid_graph_trainer_matches_eager: a failed
hipGraph capture aborts inside the HIP runtime, so the comparison runs out of process."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402  (first: it sets the BLAS environment before torch loads; make_batches)
import torch  # noqa: E402
from taiyaki_amd import _lib, models, parallel, train  # noqa: E402


def main():
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
    tr_a = train.Trainer(net_a, parallel.FlatGradArena(net_a), clip_num_mads=0, clip_window=3)
    tr_b = train.Trainer(net_b, parallel.FlatGradArena(net_b), clip_num_mads=0, clip_window=3)
    hy = train.HybridGraphTrainer(tr_b, batches[0], seq_capacity=nbatch * (T + 1))
    hy.load(batches[0])
    hy.capture(warmup=1)        # one eager step + the capture's own tail step, both on batch 0
    la = [float(tr_a.step(batches[0])), float(tr_a.step(batches[0]))]
    worst = 0.0
    for i in range(1, 6):
        b = batches[i % 3]
        la.append(float(tr_a.step(b)))
        lb = float(hy.step(b))
        worst = max(worst, abs(la[-1] - lb) / max(1e-6, abs(la[-1])))
    torch.cuda.synchronize()
    _lib.raise_if_nonfinite()
    pw = max(float((pa - pb).abs().max()) for pa, pb in zip(net_a.parameters(), net_b.parameters()))
    print("hybrid-ok loss_rel=%.3e param_abs=%.3e losses=%s" % (worst, pw, ["%.5f" % x for x in la]))


if __name__ == "__main__":
    main()
