#!/usr/bin/env python
"""Which op breaks hipGraph capture?  Each candidate runs in its own subprocess.
    python tools/graph_probe.py            (driver)
    python tools/graph_probe.py <opname>   (single probe)
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OPS = ["linear", "linear_bwd", "addmm", "matmul", "lstm_fwd", "lstm_bwd", "conv1d", "logz", "crf", "adamw"]


def probe(name):
    import torch
    from taiyaki_amd import _lib, ctc, layers, synth
    _lib.set_strict(False)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 40).to(dev)
    lstm = torch.nn.LSTM(256, 256).to(dev)
    conv = torch.nn.Conv1d(16, 256, 19, stride=5).to(dev)
    x = torch.randn(200, 16, 256, device=dev, requires_grad=True)
    xc = torch.randn(16, 16, 1000, device=dev)
    inp = synth.crf_case(200, 16, 1)
    sc = torch.from_numpy(inp["scores"]).to(dev).requires_grad_()
    seqs = torch.from_numpy(inp["seqs"]).to(dev, torch.int32)
    seqlens = torch.from_numpy(inp["seqlens"]).to(dev, torch.int32)
    p = torch.nn.Parameter(torch.randn(1000, device=dev))
    p.grad = torch.randn(1000, device=dev)
    opt = torch.optim.AdamW([p], capturable=True)

    def run():
        if name == "linear":
            return lin(x.detach())
        if name == "linear_bwd":
            lin(x).sum().backward()
        if name == "addmm":
            return torch.addmm(lin.bias, x.detach().view(-1, 256), lin.weight.t())
        if name == "matmul":
            return x.detach().view(-1, 256) @ lin.weight.t()
        if name == "lstm_fwd":
            return lstm(x.detach())[0]
        if name == "lstm_bwd":
            lstm(x)[0].sum().backward()
        if name == "conv1d":
            return conv(xc)
        if name == "logz":
            layers.flipflop_logpartition(sc).sum().backward()
        if name == "crf":
            ctc.crf_flipflop_loss(sc, seqs, seqlens, 1.0).sum().backward()
        if name == "adamw":
            opt.step()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    torch.cuda.synchronize()
    print("PROBE-OK", name)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        probe(sys.argv[1])
    else:
        for op in OPS:
            for env_extra in ({}, {"ROCBLAS_USE_HIPBLASLT": "0", "MIOPEN_GEMM_ENFORCE_BACKEND": "1",
                                   "DISABLE_ADDMM_CUDA_LT": "1", "TORCH_BLAS_PREFER_HIPBLASLT": "0"}):
                env = dict(os.environ, **env_extra)
                r = subprocess.run([sys.executable, __file__, op], env=env, capture_output=True, text=True,
                                   timeout=300)
                ok = "PROBE-OK" in r.stdout
                err = [l for l in r.stderr.splitlines() if "amdgpu.ids" not in l][-1:] if not ok else []
                print("%-11s env=%-7s %s %s" % (op, "forced" if env_extra else "default",
                                               "ok" if ok else "FAIL", " ".join(err)[:200]), flush=True)
