#!/usr/bin/env python
"""Pack a Taiyaki mapped-signal HDF5 file (v8, docs/FILE_FORMATS.md:43-75) into the .npz that
`taiyaki_amd.mapped_signal.MappedSignalStore.from_npz` loads.  Needs h5py (not in the build image:
run it wherever the training data lives).

    python tools/mapped_signal_to_npz.py mapped_reads.hdf5 mapped_reads.npz [--limit N]
File layouts read here: attributes `alphabet`, `collapse_alphabet` on the root; per-read: group
`Reads/<read_id>` with datasets `Dacs` (int16), `Ref_to_signal` (int32), `Reference` (int16) and
attributes `shift_frompA`, `scale_frompA`, `range`, `offset`, `digitisation`; batched (the writers'
default): groups `Batches/Batch_<k>` with the concatenated arrays and their `_lengths`.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_hdf5(path, limit=None):
    import h5py
    reads = []
    with h5py.File(path, "r") as h5:
        def attr(obj, name, default=None):
            v = obj.attrs.get(name, default)
            return v.decode() if isinstance(v, bytes) else v
        alphabet = attr(h5, "alphabet", "ACGT")
        collapse = attr(h5, "collapse_alphabet", alphabet)
        if "Reads" not in h5 and "Batches" in h5:
            # the writers' default layout (BatchHDF5Writer, mapped_signal_files.py:562-668): the same
            # splitting as the built-in reader, over h5py datasets
            from taiyaki_amd import hdf5_lite

            class _Ds:
                def __init__(self, d):
                    self.d = d

                def read(self):
                    return self.d[()]

            class _Grp:
                def __init__(self, g):
                    self.g = g

                def keys(self):
                    return list(self.g.keys())

                def __getitem__(self, k):
                    v = self.g[k]
                    return _Grp(v) if isinstance(v, h5py.Group) else _Ds(v)

            return hdf5_lite.reads_of_batches(_Grp(h5["Batches"]), limit), str(alphabet), str(collapse)
        for k, rid in enumerate(h5["Reads"]):
            if limit is not None and k >= limit:
                break
            g = h5["Reads"][rid]
            reads.append(dict(read_id=rid, Dacs=g["Dacs"][()], Ref_to_signal=g["Ref_to_signal"][()],
                              Reference=g["Reference"][()],
                              **{a: float(g.attrs[a]) for a in ("shift_frompA", "scale_frompA", "range",
                                                                "offset", "digitisation")}))
    return reads, str(alphabet), str(collapse)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("hdf5")
    ap.add_argument("npz")
    ap.add_argument("--limit", type=int, default=None)
    args = ap.parse_args()
    from taiyaki_amd import mapped_signal
    reads, alphabet, collapse = read_hdf5(args.hdf5, args.limit)
    mapped_signal.save_npz(args.npz, reads, alphabet, collapse)
    print("packed %d reads, %d samples -> %s" % (len(reads), sum(len(r["Dacs"]) for r in reads), args.npz))


if __name__ == "__main__":
    main()
