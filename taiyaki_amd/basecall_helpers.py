"""Chunk stitching around the decode kernels (``taiyaki/basecall_helpers.py:46-94``)."""
import torch


def stitch_chunks(out, chunk_starts, chunk_ends, stride, path_stitching=False):
    """Join per-chunk network outputs / Viterbi paths / error probabilities of overlapping
    chunks: every overlap is cut at its midpoint.  `out` is (time, chunks, ...); returns
    (blocks, ...).  `path_stitching` shifts every cut by one (the path has one more row than
    the blocks; the caller of the reference passes False, basecall.py:222-231)."""
    nchunks = out.shape[1]
    if nchunks == 1:
        return out[:, 0]
    shift = 1 if path_stitching else 0
    pieces = []
    for i in range(nchunks):
        if i == 0:
            lo = chunk_starts[0] // stride
        else:
            lo = (chunk_ends[i - 1] - chunk_starts[i]) // (2 * stride) + shift
        if i == nchunks - 1:
            hi = (chunk_ends[i] - chunk_starts[i]) // stride + shift
        elif i == 0:
            hi = (chunk_ends[0] + chunk_starts[1]) // (2 * stride) + shift
        else:
            hi = (chunk_ends[i] + chunk_starts[i + 1] - 2 * chunk_starts[i]) // (2 * stride) + shift
        pieces.append(out[lo:hi, i])
    return torch.cat(pieces, 0)
