#!/usr/bin/env python
"""What would folding GlobalNormFlipFlop's `5 tanh` (layers.py:1545-1551) and its backward into the
loss kernels save?  Times exactly the elementwise kernels the fold would remove -- y = 5 tanh(x)
forward, dx = dy 5 (1 - tanh^2) backward -- at the train step's shape inside a replayed hipGraph
(the way the step runs them), against the loss path they would be folded into."""
import argparse
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=800)
    ap.add_argument("--N", type=int, default=128)
    ap.add_argument("--S", type=int, default=40)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    x = torch.randn(args.T, args.N, args.S, device=dev, requires_grad=True)
    gy = torch.randn(args.T, args.N, args.S, device=dev)

    def fwd_bwd():
        y = 5.0 * torch.tanh(x)
        (gx,) = torch.autograd.grad(y, x, gy)
        return gx

    def fwd_only():
        with torch.no_grad():
            return 5.0 * torch.tanh(x)

    out = {}
    for name, fn in (("forward", fwd_only), ("forward+backward", fwd_bwd)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(10):
                    fn()
        torch.cuda.synchronize()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps // 10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) * 1e3 / (args.reps // 10 * 10)
    mb = args.T * args.N * args.S * 4 / 1e6
    print("5 tanh at T=%d N=%d S=%d (%.1f MB): forward %.1f us, forward + backward %.1f us per step (graph replay)"
          % (args.T, args.N, args.S, mb, out["forward"], out["forward+backward"]))


if __name__ == "__main__":
    main()
