"""Device-resident mapped-signal training set and chunk batches.

Host-side mirror of the reference's chunk producer -- ``chunk_selection.sample_chunks`` /
``sample_filter_parameters`` (chunk_selection.py:29-131) on top of
``SignalMapping.get_chunk_with_sample_length`` and ``Chunk.apply_filters``
(signal_mapping.py:515-557, 676-716), and the stacking / flip-flop coding of
``prepare_random_batches`` (bin/train_flipflop.py:78-142) -- with the work done by the gfx950
kernels of csrc/chunk_kernels.hip through the C ABI.  The reads (the per-read dictionaries of
``SignalMapping.get_read_dictionary``, signal_mapping.py:318-350) are packed ONCE into device
memory; a batch is three small launches on the current stream and never touches the host: what
comes back is the (chunk_len, nbatch, 1) float32 signal tensor and the concatenated flip-flop
coded sequences the loss operators of ``taiyaki_amd.ctc`` take.

Two ways to draw (read, start) candidates:
  * ``reference_candidates``: numpy's generator called in exactly the reference's order
    (``randint(nreads)``, then ``randint(spare_length)`` when the read is long enough), so a
    seeded run selects the chunks the reference selects -- this is what the parity tests pin;
  * on the device (``torch.randint`` / ``torch.rand``): no host work at all.

Reading the HDF5 container itself (mapped_signal_files.py) needs h5py, which this image does
not have: ``MappedSignalStore`` takes read dictionaries from any loader.
"""
import ctypes
from collections import namedtuple

import numpy as np
import torch

from taiyaki_amd import _lib
from taiyaki_amd.clipping import med_mad

REASONS = ("pass", "emptysequence", "emptysignal", "tooshort", "nullmapping", "pathbuffer",
           "meandwell", "maxdwell")       # signal_mapping.py:611-623, codes of chunk_kernels.hip
_TINY = 0.00000001                        # signal_mapping.py:609


class FILTER_PARAMETERS(namedtuple('FILTER_PARAMETERS', (
        'filter_mean_dwell', 'filter_max_dwell', 'filter_min_pass_fraction',
        'median_meandwell', 'mad_meandwell', 'model_stride', 'path_buffer'))):
    """chunk_selection.py:9-26 (same fields, same order)."""


class _Store(ctypes.Structure):           # tk_mapped_store
    _fields_ = [("dacs", ctypes.c_void_p), ("dacs_off", ctypes.c_void_p),
                ("ref_to_signal", ctypes.c_void_p), ("rts_off", ctypes.c_void_p),
                ("reference", ctypes.c_void_p), ("scaling", ctypes.c_void_p),
                ("mapped", ctypes.c_void_p), ("nreads", ctypes.c_size_t)]


class _Filter(ctypes.Structure):          # tk_chunk_filter
    _fields_ = [("enabled", ctypes.c_int), ("model_stride", ctypes.c_int),
                ("filter_mean_dwell", ctypes.c_double), ("filter_max_dwell", ctypes.c_double),
                ("median_meandwell", ctypes.c_double), ("mad_meandwell", ctypes.c_double),
                ("path_buffer", ctypes.c_double)]


def _filter_struct(fp):
    on = not (fp.median_meandwell is None or fp.mad_meandwell is None or
              fp.model_stride is None or fp.path_buffer is None)       # signal_mapping.py:688-695
    if not on:
        return _Filter(0, 1, 0.0, 0.0, 0.0, 0.0, 0.0)
    return _Filter(1, int(fp.model_stride), float(fp.filter_mean_dwell), float(fp.filter_max_dwell),
                   float(fp.median_meandwell), float(fp.mad_meandwell), float(fp.path_buffer))


def mapped_dacs_regions(reads):
    """get_mapped_dacs_region (signal_mapping.py:366-380) of every read: first / last
    Ref_to_signal value in [0, siglen]; (0, 0) when nothing is mapped.  (nreads, 2) int32."""
    mapped = np.zeros((len(reads), 2), dtype=np.int32)
    for i, r in enumerate(reads):
        a = np.asarray(r["Ref_to_signal"])
        v = a[(a >= 0) & (a <= len(r["Dacs"]))]
        if len(v):
            mapped[i] = v[0], v[-1]
    return mapped


def draw_reference_candidates(mapped, attempts, chunk_len, rng=np.random, select_strands_randomly=True,
                              first_strand_index=0):
    """(read_number, start_sample) for `attempts` tries, drawing from `rng` in the reference's
    order: chunk_selection.py:78-80 picks the read, signal_mapping.py:534-545 draws the start
    only when the read has spare length.  Host arrays (int32)."""
    nreads = len(mapped)
    reads = np.empty(attempts, dtype=np.int32)
    starts = np.zeros(attempts, dtype=np.int32)
    for k in range(attempts):
        rn = rng.randint(nreads) if select_strands_randomly else (first_strand_index + k) % nreads
        reads[k] = rn
        spare = int(mapped[rn, 1]) - int(mapped[rn, 0]) - chunk_len
        if spare > 0:
            starts[k] = rng.randint(spare)
    return reads, starts


# ---- packed on-disk form --------------------------------------------------------------------
# The arrays of tk_mapped_store as one .npz: what `MappedSignalStore` uploads, so a training set is
# packed once (tools/mapped_signal_to_npz.py does it from the reference's HDF5 container on a
# machine that has h5py) and then loaded with a few large reads instead of one HDF5 group per read.
_NPZ_KEYS = ("dacs", "dacs_off", "ref_to_signal", "rts_off", "reference", "scaling", "read_ids",
             "alphabet", "collapse_alphabet")


def pack_reads(reads, alphabet="ACGT", collapse_alphabet=None):
    """Read dictionaries (signal_mapping.py:318-350) -> dict of concatenated arrays (_NPZ_KEYS)."""
    dacs = [np.ascontiguousarray(r["Dacs"], dtype=np.int16) for r in reads]
    rts = [np.ascontiguousarray(r["Ref_to_signal"], dtype=np.int32) for r in reads]
    ref = [np.ascontiguousarray(r["Reference"], dtype=np.int16) for r in reads]
    return dict(
        dacs=np.concatenate(dacs) if dacs else np.zeros(0, np.int16),
        dacs_off=np.concatenate([[0], np.cumsum([len(d) for d in dacs])]).astype(np.int64),
        ref_to_signal=np.concatenate(rts) if rts else np.zeros(0, np.int32),
        rts_off=np.concatenate([[0], np.cumsum([len(a) for a in rts])]).astype(np.int64),
        reference=np.concatenate(ref) if ref else np.zeros(0, np.int16),
        scaling=np.array([[r["offset"], r["range"], r["digitisation"], r["shift_frompA"], r["scale_frompA"]]
                          for r in reads], dtype=np.float64).reshape(len(reads), 5),
        read_ids=np.array([str(r.get("read_id", i)) for i, r in enumerate(reads)]),
        alphabet=np.array(alphabet), collapse_alphabet=np.array(collapse_alphabet or alphabet))


def save_npz(path, reads, alphabet="ACGT", collapse_alphabet=None):
    np.savez(path, **pack_reads(reads, alphabet, collapse_alphabet))


def load_npz(path):
    """-> (read dictionaries (views into the packed arrays), alphabet, collapse_alphabet)."""
    z = np.load(path, allow_pickle=False)
    missing = [k for k in _NPZ_KEYS if k not in z.files]
    if missing:
        raise ValueError("%s is not a packed mapped-signal file: missing %s" % (path, missing))
    d, do, t, to, f, sc = (z[k] for k in ("dacs", "dacs_off", "ref_to_signal", "rts_off", "reference", "scaling"))
    reads = []
    for i, rid in enumerate(z["read_ids"]):
        reads.append(dict(read_id=str(rid), Dacs=d[do[i]:do[i + 1]], Ref_to_signal=t[to[i]:to[i + 1]],
                          Reference=f[to[i] - i:to[i + 1] - i - 1], offset=float(sc[i, 0]), range=float(sc[i, 1]),
                          digitisation=float(sc[i, 2]), shift_frompA=float(sc[i, 3]),
                          scale_frompA=float(sc[i, 4])))
    return reads, str(z["alphabet"]), str(z["collapse_alphabet"])


class ChunkBatch:
    """One sampled batch, all tensors on the device.  ``indata`` (chunk_len, nwant, 1) float32,
    ``seqs`` int32 (capacity; the first ``seqoff[-1]`` entries are valid), ``seqlens`` (nwant)
    int32, ``seqoff`` (nwant + 1) int64, ``mod_cats`` or None.  Columns beyond the accepted count
    (too few candidates passed the filters) are zero with seqlen 0; ``trimmed()`` drops them the
    way the reference's shorter batch does (one host sync)."""

    def __init__(self, indata, seqs, seqlens, seqoff, mod_cats, counts, sel, cand_read, dacstart,
                 seqlen_cand, status):
        self.indata, self.seqs, self.seqlens, self.seqoff = indata, seqs, seqlens, seqoff
        self.mod_cats, self.counts, self.sel = mod_cats, counts, sel
        self.cand_read, self.dacstart, self.seqlen_cand, self.status = cand_read, dacstart, seqlen_cand, status

    def _counts(self):
        c = self.counts.cpu().numpy()
        if int(self.status.item()) & 4:
            raise RuntimeError("chunk batch: sequence buffer too small (raise max_bases_per_chunk)")
        return c

    @property
    def naccepted(self):
        return int(self._counts()[len(REASONS)])

    @property
    def attempts(self):
        return int(self._counts()[len(REASONS) + 1])

    def rejections(self):
        """{reason: count} over the attempts made, like sample_chunks' second return value."""
        c = self._counts()
        return {k: int(c[i]) for i, k in enumerate(REASONS) if c[i]}

    def trimmed(self):
        """(indata, seqs, seqlens, mod_cats) with only the accepted chunks: the tuple
        prepare_random_batches yields (bin/train_flipflop.py:142)."""
        n = self.naccepted
        total = int(self.seqoff[n].item())
        return (self.indata[:, :n].contiguous(), self.seqs[:total], self.seqlens[:n],
                None if self.mod_cats is None else self.mod_cats[:total])


class MappedSignalStore:
    """Reads packed into device memory (see include/taiyaki_amd_flipflop.h tk_mapped_store)."""

    def __init__(self, reads, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("MappedSignalStore lives in GPU memory; device=%s (no CPU fallback)" % device)
        if len(reads) == 0:
            raise ValueError("no reads")
        self.device = device
        self.read_ids = [r.get("read_id", str(i)) for i, r in enumerate(reads)]
        dacs = [np.ascontiguousarray(r["Dacs"], dtype=np.int16) for r in reads]
        rts = [np.ascontiguousarray(r["Ref_to_signal"], dtype=np.int32) for r in reads]
        ref = [np.ascontiguousarray(r["Reference"], dtype=np.int16) for r in reads]
        for i, (a, b) in enumerate(zip(rts, ref)):
            if len(a) != len(b) + 1:            # SignalMapping.check, signal_mapping.py:101-106
                raise ValueError("read %s: Ref_to_signal must be one longer than Reference" % self.read_ids[i])
            if np.any(np.diff(a) < 0):
                raise ValueError("read %s: mapping does not increase monotonically" % self.read_ids[i])
        self.dacs_off = np.concatenate([[0], np.cumsum([len(d) for d in dacs])]).astype(np.int64)
        self.rts_off = np.concatenate([[0], np.cumsum([len(a) for a in rts])]).astype(np.int64)
        self.mapped = mapped_dacs_regions(reads)
        mapped = self.mapped
        scaling = np.array([[r["offset"], r["range"], r["digitisation"], r["shift_frompA"],
                             r["scale_frompA"]] for r in reads], dtype=np.float64)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)     # noqa: E731
        self._t = dict(dacs=up(np.concatenate(dacs)), dacs_off=up(self.dacs_off),
                       rts=up(np.concatenate(rts)), rts_off=up(self.rts_off),
                       ref=up(np.concatenate(ref) if sum(len(b) for b in ref) else np.zeros(1, np.int16)),
                       scaling=up(scaling), mapped=up(mapped))
        t = self._t
        self._struct = _Store(t["dacs"].data_ptr(), t["dacs_off"].data_ptr(), t["rts"].data_ptr(),
                              t["rts_off"].data_ptr(), t["ref"].data_ptr(), t["scaling"].data_ptr(),
                              t["mapped"].data_ptr(), len(reads))

    @classmethod
    def from_npz(cls, path, device):
        """A training set packed by `save_npz` / tools/mapped_signal_to_npz.py."""
        reads, alphabet, collapse = load_npz(path)
        store = cls(reads, device)
        store.alphabet, store.collapse_alphabet = alphabet, collapse
        return store

    @classmethod
    def from_hdf5(cls, path, device, limit=None):
        """A mapped-signal HDF5 file as the reference's MappedSignalReader reads it
        (mapped_signal_files.py:262-350, docs/FILE_FORMATS.md:43-75), through the built-in
        parser `hdf5_lite` (no h5py needed): the classic layout and the HDF5 1.8 layout the per-read
        writer asks for (libver='v108'), per-read groups or the batch layout."""
        from taiyaki_amd import hdf5_lite
        info, reads = hdf5_lite.read_mapped_signal_file(path, limit=limit)
        store = cls(reads, device)
        store.alphabet, store.collapse_alphabet = info["alphabet"], info["collapse_alphabet"]
        store.mod_long_names = info["mod_long_names"]
        return store

    @property
    def nreads(self):
        return len(self.read_ids)

    @property
    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in self._t.values())

    # ---- candidates ------------------------------------------------------------------
    def reference_candidates(self, attempts, chunk_len, rng=np.random, select_strands_randomly=True,
                             first_strand_index=0):
        """See draw_reference_candidates."""
        return draw_reference_candidates(self.mapped, attempts, chunk_len, rng, select_strands_randomly,
                                         first_strand_index)

    # ---- the three launches ----------------------------------------------------------
    def _locate(self, cand_read, cand_start, cand_frac, chunk_len, fp):
        L = _lib.lib()
        n = cand_read.numel()
        dev = self.device
        out = dict(reason=torch.empty(n, dtype=torch.uint8, device=dev),
                   dacstart=torch.empty(n, dtype=torch.int32, device=dev),
                   seqstart=torch.empty(n, dtype=torch.int32, device=dev),
                   seqlen=torch.empty(n, dtype=torch.int32, device=dev),
                   maxdwell=torch.empty(n, dtype=torch.int32, device=dev))
        filt = _filter_struct(fp)
        rc = L.tk_chunks_locate_dev(ctypes.byref(self._struct), _lib.ptr(cand_read), _lib.ptr(cand_start),
                                    _lib.ptr(cand_frac), n, chunk_len, ctypes.byref(filt),
                                    _lib.ptr(out["reason"]), _lib.ptr(out["dacstart"]),
                                    _lib.ptr(out["seqstart"]), _lib.ptr(out["seqlen"]),
                                    _lib.ptr(out["maxdwell"]), _lib.stream_ptr())
        _lib.check(rc, "tk_chunks_locate_dev")
        return out

    def _select(self, loc, nwant):
        L = _lib.lib()
        dev = self.device
        sel = torch.empty(max(nwant, 1), dtype=torch.int32, device=dev)
        seqoff = torch.empty(nwant + 1, dtype=torch.int64, device=dev)
        counts = torch.empty(len(REASONS) + 2, dtype=torch.int32, device=dev)
        rc = L.tk_chunks_select_dev(_lib.ptr(loc["reason"]), _lib.ptr(loc["seqlen"]), loc["reason"].numel(),
                                    nwant, _lib.ptr(sel), _lib.ptr(seqoff), _lib.ptr(counts),
                                    _lib.stream_ptr())
        _lib.check(rc, "tk_chunks_select_dev")
        return sel, seqoff, counts

    def _candidates(self, attempts, chunk_len, candidates, rng, select_strands_randomly, first_strand_index):
        dev = self.device
        if candidates is not None:
            cr, cs = candidates
        elif rng is not None or not select_strands_randomly:
            cr, cs = self.reference_candidates(attempts, chunk_len, rng if rng is not None else np.random,
                                               select_strands_randomly, first_strand_index)
        else:       # drawn on the device
            cr = torch.randint(self.nreads, (attempts,), device=dev, dtype=torch.int32)
            return cr, None, torch.rand(attempts, device=dev, dtype=torch.float64)
        as_dev = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32).to(dev).contiguous()   # noqa: E731
        return as_dev(cr), as_dev(cs), None

    def sample_chunks(self, number_to_sample, chunk_len, filter_params, standardize=True,
                      select_strands_randomly=True, first_strand_index=0, *, reverse=False, ncan=4,
                      can_labels=None, mod_labels=None, candidates=None, rng=None,
                      max_bases_per_chunk=None):
        """chunk_selection.sample_chunks (chunk_len in samples) + the batch assembly of
        prepare_random_batches.  number_to_sample None / 0 = one chunk per read.

        candidates: explicit (read_numbers, start_samples); rng: a numpy generator to draw from in
        the reference's order (np.random reproduces a seeded reference run); neither: drawn on the
        device.  max_bases_per_chunk sizes the sequence buffer (default chunk_len: every base of a
        chunk that passes the path-buffer filter spans at least one sample)."""
        with torch.cuda.device(self.device):
            nwant = self.nreads if not number_to_sample else int(number_to_sample)
            attempts = int(nwant / filter_params.filter_min_pass_fraction)      # chunk_selection.py:71-72
            cr, cs, cf = self._candidates(attempts, chunk_len, candidates, rng, select_strands_randomly,
                                          first_strand_index)
            loc = self._locate(cr, cs, cf, chunk_len, filter_params)
            sel, seqoff, counts = self._select(loc, nwant)
            dev = self.device
            cap = nwant * int(max_bases_per_chunk if max_bases_per_chunk else max(chunk_len, 1))
            indata = torch.empty((chunk_len, nwant, 1), dtype=torch.float32, device=dev)
            seqs = torch.zeros(max(cap, 1), dtype=torch.int32, device=dev)       # tail beyond seqoff[-1] stays 0
            seqlens = torch.empty(max(nwant, 1), dtype=torch.int32, device=dev)[:nwant]
            # strict mode: a word of its own, read by ChunkBatch._counts(); deferred mode (training
            # loops): the process-wide word that `_lib.raise_if_nonfinite()` checks once per step, so a
            # too small `max_bases_per_chunk` cannot silently truncate `seqs`
            status = _lib.status_word(dev)
            cl = ml = mc = None
            if mod_labels is not None:
                cl = torch.as_tensor(np.asarray(can_labels), dtype=torch.int32).to(dev)
                ml = torch.as_tensor(np.asarray(mod_labels), dtype=torch.int32).to(dev)
                mc = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
            rc = _lib.lib().tk_chunks_gather_dev(
                ctypes.byref(self._struct), _lib.ptr(cr), _lib.ptr(loc["dacstart"]), _lib.ptr(loc["seqstart"]),
                _lib.ptr(loc["seqlen"]), _lib.ptr(sel), _lib.ptr(seqoff), _lib.ptr(counts), nwant, chunk_len,
                int(bool(reverse)), int(bool(standardize)), ncan, _lib.ptr(cl), _lib.ptr(ml),
                _lib.ptr(indata), _lib.ptr(seqs), cap, _lib.ptr(seqlens), _lib.ptr(mc), _lib.ptr(status),
                _lib.stream_ptr())
            _lib.check(rc, "tk_chunks_gather_dev")
            # the host knows a bound on every sequence of the batch without looking at it: a chunk
            # that passed the path-buffer filter has chunk_len / (L * stride) > path_buffer
            # (signal_mapping.py:699-703).  The tensor carries it (ctc.set_max_seqlen): the CRF launch
            # is sized by it and a captured step can check the batch against its capacity, no sync
            if filter_params.model_stride is not None and filter_params.path_buffer is not None \
                    and filter_params.median_meandwell is not None and filter_params.mad_meandwell is not None:
                bound = int(chunk_len / (float(filter_params.model_stride) * float(filter_params.path_buffer))) + 1
                seqlens.tk_max_seqlen = min(bound, max(int(chunk_len), 1))
            return ChunkBatch(indata, seqs, seqlens, seqoff, mc, counts, sel, cr, loc["dacstart"],
                              loc["seqlen"], status)

    def sample_filter_parameters(self, number_to_sample, chunk_len, filter_mean_dwell, filter_max_dwell,
                                 filter_min_pass_fraction, model_stride, path_buffer, *, candidates=None,
                                 rng=None):
        """chunk_selection.sample_filter_parameters (chunk_selection.py:98-131): median and
        scaled MAD of the mean dwell of unfiltered chunks."""
        nofilter = FILTER_PARAMETERS(filter_mean_dwell, filter_max_dwell, filter_min_pass_fraction,
                                     None, None, None, None)
        with torch.cuda.device(self.device):
            nwant = self.nreads if not number_to_sample else int(number_to_sample)
            attempts = int(nwant / filter_min_pass_fraction)
            cr, cs, cf = self._candidates(attempts, chunk_len, candidates, rng, True, 0)
            loc = self._locate(cr, cs, cf, chunk_len, nofilter)
            sel, _, counts = self._select(loc, nwant)
            n = int(counts[len(REASONS)].item())
            L = loc["seqlen"][sel[:n].long()].cpu().numpy()
        meandwells = [chunk_len / (int(x) + _TINY) for x in L]        # Chunk.mean_dwell, :652-658
        med, mad = med_mad(meandwells)
        return FILTER_PARAMETERS(filter_mean_dwell, filter_max_dwell, filter_min_pass_fraction,
                                 med, mad, model_stride, path_buffer)
