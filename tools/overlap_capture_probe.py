#!/usr/bin/env python
"""Lab: the fused loss with kernel B on the side queue (tk_flipflop_loss_overlap) -- eager in both forms, and
captured into a torch.cuda.graph with mode 2 (round 3: the capture crashed inside the HIP runtime).  Run under
`timeout`; prints what happened at each stage.

    timeout 120 python tools/overlap_capture_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from taiyaki_amd import _lib, ctc  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    T, N, S = 800, 128, 40
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(T, N, S, generator=g).to(dev)
    lens = torch.randint(200, 500, (N,), generator=g, dtype=torch.int32)
    seqs = torch.randint(0, 4, (int(lens.sum()),), generator=g, dtype=torch.int32)
    lens.tk_max_seqlen = 512

    def run():
        return ctc._run_fused(x, seqs, lens, 1.0, True)

    out = {}
    for mode in (0, 1):
        L.tk_flipflop_loss_overlap(mode)
        lv, gr, lz = run()
        torch.cuda.synchronize()
        out[mode] = (lv.clone(), gr.clone(), lz.clone())
        print("eager mode %d: loss %.6f  grad abs sum %.4f" % (mode, float(lv.mean()), float(gr.abs().sum())), flush=True)
    dl = float((out[0][0] - out[1][0]).abs().max())
    dg = float((out[0][1] - out[1][1]).abs().max()) * T
    print("two forms: max |d loss| %.3g, max |d grad| x T %.3g" % (dl, dg), flush=True)
    L.tk_flipflop_loss_overlap(2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    _lib.set_strict(False)
    print("capturing with mode 2 ...", flush=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        lv, gr, lz = run()
    print("captured; replaying ...", flush=True)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    print("replayed: max |d loss| vs eager %.3g, max |d grad| x T %.3g" % (
        float((lv - out[1][0]).abs().max()), float((gr - out[1][1]).abs().max()) * T), flush=True)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for mode, label in ((2, "captured two-queue"),):
        st.record()
        for _ in range(50):
            graph.replay()
        en.record()
        torch.cuda.synchronize()
        print("%s replay: %.1f us (includes build_indices)" % (label, st.elapsed_time(en) * 1e3 / 50), flush=True)
    L.tk_flipflop_loss_overlap(1)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        run()
    st.record()
    for _ in range(50):
        g1.replay()
    en.record()
    torch.cuda.synchronize()
    print("captured one-queue replay: %.1f us" % (st.elapsed_time(en) * 1e3 / 50), flush=True)


if __name__ == "__main__":
    main()
