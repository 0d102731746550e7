#!/usr/bin/env python
"""Viterbi op timing (HIP events, C entry point, buffers allocated once): path only and full outputs, the three-wave
kernel against the one-wave kernel of rounds 1-4 (lab build, TK_VIT_V1=1).
    python tools/vitbench.py [T N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from taiyaki_amd import _lib, synth  # noqa: E402


def main():
    _lib.use_lab(True)
    dev = torch.device("cuda:0")
    shapes = [(800, 128), (4000, 256), (1600, 64), (4000, 1024), (4000, 4096)]
    if len(sys.argv) > 2:
        a = [int(x) for x in sys.argv[1:]]
        shapes = list(zip(a[::2], a[1::2]))
    L, p = _lib.lib(), _lib.ptr
    for T, N in shapes:
        x = torch.from_numpy(synth.scores(T, N, 40, 3)).to(dev)
        fwd = torch.empty(T + 1, N, 8, dtype=torch.float32, device=dev)
        tb = torch.empty(T, N, 8, dtype=torch.int64, device=dev)
        path = torch.empty(T + 1, N, dtype=torch.int64, device=dev)
        wsb = L.tk_flipflop_viterbi_workspace_bytes(T, N, 4)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        out = {}
        for v1 in ("0", "1"):
            os.environ["TK_VIT_V1"] = v1
            full = lambda: _lib.check(L.tk_flipflop_viterbi_dev(p(x), T, N, 4, p(fwd), p(tb), p(path), p(ws), wsb, _lib.stream_ptr()), "v")  # noqa: E731
            ponly = lambda: _lib.check(L.tk_flipflop_viterbi_dev(p(x), T, N, 4, None, None, p(path), p(ws), wsb, _lib.stream_ptr()), "v")  # noqa: E731
            reps = 20 if T * N < 2e6 else 8
            pm, pmin = bench._events_mean_min(ponly, reps, warm=3)
            fm, fmin = bench._events_mean_min(full, reps, warm=3)
            out[v1] = (pm, pmin, fm, fmin, path.clone(), fwd.clone(), tb.clone())
        same = all(torch.equal(a, b) for a, b in zip(out["0"][4:], out["1"][4:]))
        alg = T * N * 160 + (T + 1) * N * 8
        print("T=%d N=%d  three waves: path only %.1f us (min %.1f) = %.3f of 8 TB/s, full %.1f us | one wave: path only %.1f, "
              "full %.1f | outputs identical: %s" % (T, N, out["0"][0] * 1e6, out["0"][1] * 1e6, alg / out["0"][0] / 8e12,
                                                    out["0"][2] * 1e6, out["1"][0] * 1e6, out["1"][2] * 1e6, same), flush=True)


if __name__ == "__main__":
    main()
