#!/usr/bin/env python
"""Pack a Taiyaki mapped-signal HDF5 file (v8, docs/FILE_FORMATS.md:43-75) into the .npz that
`taiyaki_amd.mapped_signal.MappedSignalStore.from_npz` loads.  Needs h5py (not in the build image:
run it wherever the training data lives).

    python tools/mapped_signal_to_npz.py mapped_reads.hdf5 mapped_reads.npz [--limit N]
File layout read here: attributes `alphabet`, `collapse_alphabet` on the root; group `Reads/<read_id>`
with datasets `Dacs` (int16), `Ref_to_signal` (int32), `Reference` (int16) and attributes
`shift_frompA`, `scale_frompA`, `range`, `offset`, `digitisation`.
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
