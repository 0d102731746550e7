"""Drop-in for the hot-path entry points of ``taiyaki.layers`` plus the layer stack
that feeds them.

* ``flipflop_logpartition`` (taiyaki/layers.py:1875-1890) runs the gfx950 HIP
  kernels (csrc/logz_kernels.hip) instead of cupy / the T-step torch loop.
* ``Convolution``, ``Lstm``, ``GruMod``, ``Reverse``, ``Serial``, ``GlobalNormFlipFlop``
  and ``GlobalNormFlipFlopCatMod`` restate the reference layers minimally on
  PyTorch-ROCm (MIOpen LSTM/GRU, Conv1d): by scope they STAY PyTorch
  (BASELINE.json north_star) and only produce the (T, N, S) score tensor.
"""
import numpy as np
import torch
from torch import nn

from taiyaki_amd import _lib, flipflopfings


# ---------------------------------------------------------------------------
# (B) log-partition operator
# ---------------------------------------------------------------------------
def _logz_launch(x, want_grad):
    _lib.require_gpu(x, "flipflop_logpartition")
    L = _lib.lib()
    sc = x.detach().float().contiguous()
    if sc.data_ptr() % 16 != 0:
        sc = sc.clone()
    T, N, S = sc.shape
    nbase = flipflopfings.nbase_flipflop(S)
    dev = sc.device
    with torch.cuda.device(dev):
        logz = torch.empty(N, dtype=torch.float32, device=dev)
        grad = torch.empty_like(sc) if want_grad else None
        wsb = L.tk_flipflop_logz_workspace_bytes(T, N, nbase)
        if wsb == 0:
            raise RuntimeError("flipflop_logpartition: nbase=%d is not built" % nbase)
        ws = _lib.workspace(wsb, dev, "logz")
        status = _lib.status_word(dev)
        rc = L.tk_flipflop_logz_dev(_lib.ptr(sc), T, N, nbase, _lib.ptr(logz), _lib.ptr(grad),
                                    _lib.ptr(ws), wsb, _lib.ptr(status), _lib.stream_ptr())
        _lib.check(rc, "tk_flipflop_logz_dev")
        _lib.finish(status)
    return logz, grad


class LogZ(torch.autograd.Function):
    """cupy_extensions/flipflop.py:338-354: logZ forward; backward =
    posterior transition probabilities * g[:, None]."""

    @staticmethod
    def forward(ctx, scores):
        logz, grad = _logz_launch(scores, ctx.needs_input_grad[0])
        if grad is not None:
            ctx.save_for_backward(grad)
        return logz

    @staticmethod
    def backward(ctx, g):
        trans, = ctx.saved_tensors
        return trans * g[:, None]


def flipflop_logpartition(x, _never_use_cupy=False):
    """layers.py:1875-1890: log-partition function for each batch element, (N,)."""
    del _never_use_cupy
    return LogZ.apply(x)


def log_partition_flipflop(scores):
    """layers.py:1277-1299, the reference's torch statement of the same quantity: (N, 1).  Here it
    is the same HIP operator (differentiable through `LogZ`), not a T-step loop."""
    return flipflop_logpartition(scores).unsqueeze(1)


def global_norm_flipflop(scores):
    """layers.py:1302-1313"""
    T = scores.shape[0]
    return scores - flipflop_logpartition(scores)[None, :, None] / np.float32(T)


# ---------------------------------------------------------------------------
# layer stack (stays PyTorch-ROCm)
# ---------------------------------------------------------------------------
def swish(x):
    return x * torch.sigmoid(x)


def _orthonormal_(param):
    """layers.py:37-96 initialises matrix parameters orthonormally."""
    with torch.no_grad():
        flat = param.view(param.shape[0], -1)
        nn.init.orthogonal_(flat)


def _truncated_normal_(param, sd):
    """layers.py:99-114"""
    with torch.no_grad():
        nn.init.trunc_normal_(param, std=sd, a=-2 * sd, b=2 * sd)


class Convolution(nn.Module):
    """layers.py:744-850: TBF in/out, pad (winlen//2, (winlen-1)//2), then activation.

    Same parameters as the reference (`conv.weight` (size, insize, winlen), `conv.bias`),
    but evaluated as window-unfold + GEMM: MIOpen's choices for these skinny 1-D
    shapes (batch 128 x 4000 samples, 1->4->16->256 channels) fall back to naive
    direct/weight-gradient kernels that cost more than the whole LSTM stack, while
    the same contraction as a rocBLAS GEMM is a few hundred microseconds.  Set
    `use_gemm=False` for the plain nn.Conv1d evaluation (identical result).
    """

    def __init__(self, insize, size, winlen, stride=1, fun=torch.tanh, use_gemm=True):
        super().__init__()
        self.winlen = winlen
        self.stride = stride
        self.use_gemm = use_gemm
        self.pad = nn.ConstantPad1d((winlen // 2, (winlen - 1) // 2), 0)
        self.conv = nn.Conv1d(insize, size, winlen, stride=stride)
        self.activation = fun
        _orthonormal_(self.conv.weight)
        _truncated_normal_(self.conv.bias, 0.5)

    def forward(self, x):
        if not self.use_gemm:
            out = self.activation(self.conv(self.pad(x.permute(1, 2, 0))))
            return out.permute(2, 0, 1)
        # x: (T, N, C) -> pad time -> windows (Tout, N, C, winlen) -> GEMM with (C*winlen, size)
        xp = nn.functional.pad(x, (0, 0, 0, 0, self.winlen // 2, (self.winlen - 1) // 2))
        win = xp.unfold(0, self.winlen, self.stride)            # (Tout, N, C, winlen) view
        tout, n = win.shape[0], win.shape[1]
        w = self.conv.weight.reshape(self.conv.weight.shape[0], -1)     # (size, C*winlen)
        out = torch.addmm(self.conv.bias, win.reshape(tout * n, -1), w.t())
        return self.activation(out.view(tout, n, -1))


class _Rnn(nn.Module):
    def __init__(self, cell):
        super().__init__()
        self.rnn = cell
        for name, param in self.rnn.named_parameters():
            if 'bias_hh' in name:       # layers.py:522-532: redundant bias frozen at zero
                param.requires_grad = False
                with torch.no_grad():
                    param.zero_()
            elif 'weight' in name:
                _orthonormal_(param)
            else:
                _truncated_normal_(param, 0.5)

    def forward(self, x):
        return self.rnn(x)[0]


class Lstm(_Rnn):
    """layers.py:491-606 (wraps nn.LSTM, bias_hh frozen)"""

    def __init__(self, insize, size):
        super().__init__(nn.LSTM(insize, size))


class GruMod(_Rnn):
    """layers.py:609-725 (wraps nn.GRU, bias_hh frozen)"""

    def __init__(self, insize, size):
        super().__init__(nn.GRU(insize, size))


class Reverse(nn.Module):
    """layers.py:117-153"""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, x):
        return torch.flip(self.layer(torch.flip(x, (0,))), (0,))


class Serial(nn.Sequential):
    """layers.py:944-982"""

    def __init__(self, layers):
        super().__init__(*layers)


class GlobalNormFlipFlop(nn.Module):
    """layers.py:1316-1411: scale * tanh(x W + b); NO normalisation inside the layer."""

    def __init__(self, insize, nbase, scale=5.0):
        super().__init__()
        self.nbase = nbase
        self.size = flipflopfings.nstate_flipflop(nbase)
        self.linear = nn.Linear(insize, self.size)
        self.scale = scale
        _orthonormal_(self.linear.weight)
        _truncated_normal_(self.linear.bias, 0.5)

    def forward(self, x):
        return self.scale * torch.tanh(self.linear(x))


class GlobalNormFlipFlopCatMod(nn.Module):
    """layers.py:1414-1640 for a canonical alphabet plus per-base modifications:
    output = [5 tanh(40 transition scores), per-base log_softmax over
    {canonical, its mods}] in the order A,(A mods),C,(C mods),...
    `can_nmods` = number of modified bases per canonical base."""

    def __init__(self, insize, can_nmods=(1, 1, 0, 0)):
        super().__init__()
        self.can_nmods = np.asarray(can_nmods)
        self.ncan_base = len(can_nmods)
        self.nmod_base = int(self.can_nmods.sum())
        self.ntrans_states = flipflopfings.nstate_flipflop(self.ncan_base)
        # layers.py:1495-1506
        self.can_mods_offsets = np.cumsum(
            np.concatenate([[0], self.can_nmods + 1])).astype(np.int32)
        self.can_indices = []
        curr = 0
        for n in self.can_nmods:
            self.can_indices.append(np.concatenate([[0], np.arange(curr + 1, curr + 1 + n)]))
            curr += n
        self.size = self.ntrans_states + 1 + self.nmod_base
        # index tensors live with the module (moved by .to(device)): nothing is uploaded in
        # forward, so the layer can be captured into a hipGraph
        for k, idx in enumerate(self.can_indices):
            self.register_buffer("can_index_%d" % k, torch.as_tensor(idx, dtype=torch.int64), persistent=False)
        self.linear = nn.Linear(insize, self.size)
        _orthonormal_(self.linear.weight)
        _truncated_normal_(self.linear.bias, 0.5)

    @property
    def nout(self):
        return self.ntrans_states + self.ncan_base + self.nmod_base

    def forward(self, x):
        y = self.linear(x)
        trans = 5.0 * torch.tanh(y[:, :, :self.ntrans_states])
        cat = y[:, :, self.ntrans_states:]
        mods = [torch.log_softmax(cat.index_select(2, getattr(self, "can_index_%d" % k)), dim=2)
                for k in range(self.ncan_base)]
        return torch.cat([trans] + mods, dim=2)
