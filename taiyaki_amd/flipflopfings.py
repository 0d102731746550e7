"""Index algebra of the flip-flop code (host side, numpy) -- same names and
semantics as taiyaki/flipflopfings.py:6-184.  The training path does NOT use
these (ids are built on the device, csrc/crf_kernels.hip); they exist so that
callers and tests written against the reference module keep working."""
import numpy as np

DEFAULT_ALPHABET = 'ACGT'


def move_indices(labels, nbase=len(DEFAULT_ALPHABET)):
    """flipflopfings.py:6-17: labels[:-1] + min(labels[1:], nbase) * 2 nbase"""
    labels = np.asarray(labels)
    return labels[:-1] + np.minimum(labels[1:], nbase) * (nbase + nbase)


def stay_indices(labels, nbase=len(DEFAULT_ALPHABET)):
    """flipflopfings.py:20-31"""
    labels = np.asarray(labels)
    return labels + np.minimum(labels, nbase) * (nbase + nbase)


def flopmask(labels):
    """flipflopfings.py:34-53: True where a label sits at an even (2nd, 4th, ...)
    position within a run of identical labels.  Vectorised: position within the run = index
    minus the index at which the run started."""
    labels = np.asarray(labels)
    n = len(labels)
    if n == 0:
        return np.zeros(0, dtype=bool)
    idx = np.arange(n)
    new_run = np.ones(n, dtype=bool)
    new_run[1:] = labels[1:] != labels[:-1]
    run_start = np.maximum.accumulate(np.where(new_run, idx, 0))
    return ((idx - run_start) & 1).astype(bool)


def flipflop_code(labels, alphabet_length=4):
    """flipflopfings.py:56-78"""
    x = np.array(labels, copy=True)
    x[flopmask(x)] += alphabet_length
    return x


def path_to_str(path, alphabet=DEFAULT_ALPHABET, include_first_source=True):
    """flipflopfings.py:81-97: collapse a flip-flop state path into a basecall."""
    path = np.asarray(path)
    move = np.ediff1d(path, to_begin=1 if include_first_source else 0) != 0
    letters = np.frombuffer((alphabet * 2).encode(), dtype='u1')
    return letters[path[move]].tobytes().decode()


def nstate_flipflop(nbase):
    """flipflopfings.py:146-168: 2 nbase (nbase + 1) transitions"""
    return 2 * nbase * (nbase + 1)


def nbase_flipflop(nstate):
    """Inverse of `nstate_flipflop` (reference: flipflopfings.py:171-184, which solves the
    quadratic in floating point and asserts an integer root).  Integer arithmetic here:
    nstate = 2 b (b + 1)  <=>  b = (isqrt(1 + 2 nstate) - 1) / 2 with an exact square."""
    nstate = int(nstate)
    root = int(np.floor(np.sqrt(1.0 + 2.0 * nstate)))
    while root * root > 1 + 2 * nstate:        # (float sqrt may land one off for large values)
        root -= 1
    while (root + 1) * (root + 1) <= 1 + 2 * nstate:
        root += 1
    nbase = (root - 1) // 2
    if nstate <= 0 or root * root != 1 + 2 * nstate or nstate_flipflop(nbase) != nstate:
        raise AssertionError("%d transitions is not a flip-flop layout: no whole number of bases b has "
                             "2 b (b + 1) = %d" % (nstate, nstate))
    return nbase
