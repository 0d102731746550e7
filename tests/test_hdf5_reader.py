"""The built-in mapped-signal HDF5 reader (taiyaki_amd/hdf5_lite.py; h5py is not in this image)
on data files the reference's own tests hold (test/data/mapped_signal_file/*.hdf5 and one raw
fast5 of test/data/reads, copied as fixtures to tests/golden/mapped_signal/).

The container is validated three ways: the documented structure and invariants of the format
(docs/FILE_FORMATS.md:43-75); an INDEPENDENT file -- the raw fast5 MinKNOW wrote for the same read
(other chunking, other filters, other writer) must hold the same samples and channel constants;
and the reference's own description of the files (2 reads from the walkthrough set,
note_on_creation_of_test_data.txt)."""
import os

import numpy as np
import pytest

from taiyaki_amd import hdf5_lite

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mapped_signal")


def test_mapped_signal_file_structure_and_invariants():
    info, reads = hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "mapped_reads_0.hdf5"))
    assert info["version"] == 8 and info["alphabet"] == "ACGT" and info["collapse_alphabet"] == "ACGT"
    assert info["mod_long_names"] == []
    assert [r["read_id"] for r in reads] == ["302c746b-1b9e-4262-af6a-859bae0c00f8",
                                             "6dc84c3b-840b-4b04-a817-013a036695f8"]
    for r in reads:
        dacs, rts, ref = r["Dacs"], r["Ref_to_signal"], r["Reference"]
        assert dacs.dtype == np.int16 and rts.dtype == np.int32 and ref.dtype == np.int16
        assert len(rts) == len(ref) + 1                         # FILE_FORMATS.md: Ref_to_signal has reflen + 1 entries
        assert np.all(np.diff(rts) >= 0) and rts[0] >= -1 and rts[-1] <= len(dacs) + 1
        assert ref.min() >= 0 and ref.max() < 4
        assert r["digitisation"] == 8192.0 and 1000 < r["range"] < 2000 and 0 < r["scale_frompA"] < 100
        # r9.4.1 DNA: about 9 samples per base
        assert 7 < (rts[-1] - rts[0]) / len(ref) < 12
    assert (len(reads[0]["Dacs"]), len(reads[0]["Reference"])) == (38344, 4090)
    assert (len(reads[1]["Dacs"]), len(reads[1]["Reference"])) == (73060, 7161)


def test_dacs_equal_the_raw_fast5_signal_of_the_same_read():
    """Independent cross-check of the chunked / deflate / shuffle decoding: the mapped-signal
    file's Dacs of read de1508c4... against the Signal dataset of that read's raw fast5."""
    info, reads = hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "mapped_remap_samref.hdf5"))
    rd = [r for r in reads if r["read_id"].startswith("de1508c4")][0]
    f5 = hdf5_lite.File(os.path.join(HERE, "de1508c4-755b-489e-9ffb-51af35c9a7e6.fast5"))
    reads_group = f5["Raw/Reads"]
    (name,) = reads_group.keys()
    sig = reads_group[name]["Signal"].read()
    assert sig.dtype == np.int16 and np.array_equal(sig, rd["Dacs"])
    assert int(reads_group[name].attrs["duration"]) == len(sig)
    ch = f5["UniqueGlobalKey/channel_id"].attrs
    assert float(ch["range"]) == rd["range"] and float(ch["offset"]) == rd["offset"]
    assert float(ch["digitisation"]) == rd["digitisation"] and ch["channel_number"] == "248"


def _generated_id(r):
    return "%08x-aaaa-4bbb-8ccc-%012x" % ((r * 2654435761) & 0xffffffff, r)


def _check_generated_read(q, r):
    """The formulas of tests/golden/mapped_signal/gen_hdf5_fixtures.c."""
    ns, nr = 40 + 3 * r, 5 + r
    assert q["read_id"] == _generated_id(r)
    np.testing.assert_array_equal(q["Dacs"], ((np.arange(ns) * 7 + r * 13) % 1000 - 300).astype(np.int16))
    np.testing.assert_array_equal(q["Reference"], ((np.arange(nr) + r) % 4).astype(np.int16))
    np.testing.assert_array_equal(q["Ref_to_signal"], (np.arange(nr + 1) * ns // nr).astype(np.int32))
    assert (q["shift_frompA"], q["scale_frompA"], q["range"], q["offset"], q["digitisation"]) == \
        (1.5 + r, 0.25 * (r + 1), 1400.0 + r, 10.0 - r, 8192.0)


def test_hdf5_18_layout_same_reads_as_the_classic_file():
    """HDF5 1.8 layout -- what the per-read writer of today asks for (libver='v108',
    mapped_signal_files.py:372): the reference's own test file re-written by the HDF5 library's
    h5repack with those bounds (superblock 2, version-2 object headers + continuation chunks, link
    messages) must give exactly the reads of the classic file."""
    f = hdf5_lite.File(os.path.join(HERE, "mapped_reads_0_v108.hdf5"))
    assert f.superblock_version == 2
    info0, reads0 = hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "mapped_reads_0.hdf5"))
    info1, reads1 = hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "mapped_reads_0_v108.hdf5"))
    assert info0 == info1 and len(reads0) == len(reads1) == 2
    by = {r["read_id"]: r for r in reads1}
    for r in reads0:
        q = by[r["read_id"]]
        assert set(q) == set(r)
        for k, v in r.items():
            if isinstance(v, np.ndarray):
                np.testing.assert_array_equal(v, q[k])
            else:
                assert v == q[k], k


def test_hdf5_18_layout_written_by_the_library_with_the_writers_options():
    """60 reads written by libhdf5 itself with PerReadHDF5Writer's options (gen_hdf5_fixtures.c: low
    bound V18, creation order tracked and indexed, gzip + shuffle chunked datasets, variable-length
    string attributes, `read_ids` in the root): the Reads group is DENSE (links in a fractal heap
    with an indirect block), every other read has ten attributes (dense attribute storage), and the
    links come back in creation order."""
    path = os.path.join(HERE, "generated_v108.hdf5")
    raw = open(path, "rb").read()
    assert raw.count(b"FRHP") > 20 and raw.count(b"FHIB") >= 1 and raw.count(b"OHDR") > 200
    f = hdf5_lite.File(path)
    assert f.superblock_version == 2 and sorted(f.keys()) == ["Reads", "read_ids"]
    assert len(f["Reads"]) == 60
    info, reads = hdf5_lite.read_mapped_signal_file(path)
    assert info["version"] == 8 and info["alphabet"] == "ACGT" and info["mod_long_names"] == []
    assert [r["read_id"] for r in reads] == [_generated_id(r) for r in range(60)]
    for r, q in enumerate(reads):
        _check_generated_read(q, r)
        if r % 2 == 0:
            assert q["mapping_score"] == 100.0 + 0.5 * r and q["mapping_method"] == "generated"
            assert len(f["Reads"][q["read_id"]].attrs) == 10
        else:
            assert "mapping_score" not in q
    ids = f["read_ids"].read()                  # (variable-length strings decode through the global heap)
    assert ids == [_generated_id(r) for r in range(60)]


def test_batch_layout_written_by_the_library():
    """BatchHDF5Writer's layout (the writers' default, mapped_signal_files.py:582-650) in a file
    written by libhdf5 itself: three batches of 10 / 10 / 3 reads, concatenated arrays with
    `_lengths`, float64 columns, a variable-length string `read_id` column, all gzip-compressed.
    Pins `reads_of_batches` on a genuine HDF5 file (round 2 could only re-pack arrays in memory)."""
    info, reads = hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "generated_batch.hdf5"))
    assert info["version"] == 8 and len(reads) == 23
    for r, q in enumerate(reads):
        _check_generated_read(q, r)
    _, few = hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "generated_batch.hdf5"), limit=12)
    assert len(few) == 12


def test_files_that_are_not_hdf5_or_too_new_are_refused(tmp_path):
    p = tmp_path / "new.hdf5"
    p.write_bytes(hdf5_lite.SIGNATURE + bytes([4, 8, 8, 0]) + bytes(64))
    with pytest.raises(hdf5_lite.Hdf5Error, match="superblock"):
        hdf5_lite.File(str(p))
    q = tmp_path / "junk.bin"
    q.write_bytes(b"not hdf5" * 100)
    with pytest.raises(hdf5_lite.Hdf5Error, match="not an HDF5 file"):
        hdf5_lite.File(str(q))


def test_chunks_from_the_real_file_match_the_oracle_sampler():
    """The reads of the real file through the numpy restatement of the reference's chunk
    sampler (oracle/chunks.py): chunks come out with the documented shapes (host side; the
    device path is compared with the same oracle in the -m gpu test below)."""
    from oracle import chunks as oc
    _, reads = hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "mapped_reads_0.hdf5"))
    rs = np.random.RandomState(3)
    cands = [(int(rs.randint(len(reads))), int(rs.randint(20000))) for _ in range(12)]
    fp = dict(filter_mean_dwell=3.0, filter_max_dwell=10.0, filter_min_pass_fraction=0.5, median_meandwell=None,
              mad_meandwell=None, model_stride=5, path_buffer=1.1)
    want, counts, attempts = oc.sample_chunks(reads, 6, 2000, fp, cands)
    assert len(want) == 6 and attempts >= 6
    indata, seqs, seqlens, _ = oc.assemble_batch(want, 4)
    assert indata.shape == (2000, 6, 1) and np.all(np.abs(indata) < 10) and np.all(seqlens > 100)


@pytest.mark.gpu
def test_store_from_hdf5_feeds_the_loss(gpu_device):
    """Real r9.4.1 reads: HDF5 file -> MappedSignalStore in HBM -> chunk batch assembled on the
    device, bit for bit what the oracle sampler assembles from the same reads and candidates ->
    HIP flip-flop loss with a finite value and gradient."""
    import torch
    from oracle import chunks as oc
    from taiyaki_amd import ctc, mapped_signal
    path = os.path.join(HERE, "mapped_reads_0.hdf5")
    store = mapped_signal.MappedSignalStore.from_hdf5(path, gpu_device)
    assert store.nreads == 2 and store.alphabet == "ACGT"
    _, reads = hdf5_lite.read_mapped_signal_file(path)
    rs = np.random.RandomState(5)
    fp = store.sample_filter_parameters(30, 2000, 3.0, 10.0, 0.5, 5, 1.1, rng=rs)
    cand = store.reference_candidates(16, 2000, rs)
    cb = store.sample_chunks(8, 2000, fp, candidates=cand)
    want, _, _ = oc.sample_chunks(reads, 8, 2000, dict(fp._asdict()), list(zip(*cand)))
    assert cb.naccepted == len(want) == 8
    indata, seqs, seqlens, _ = oc.assemble_batch(want, 4)
    got = cb.trimmed()
    assert np.array_equal(got[0].cpu().numpy().view(np.uint32), indata.view(np.uint32))
    assert np.array_equal(got[1].cpu().numpy(), seqs) and np.array_equal(got[2].cpu().numpy(), seqlens)
    T = 2000 // 5
    x = (torch.randn(T, 8, 40, device=gpu_device) * 2).requires_grad_()
    lv = ctc.flipflop_loss(x, got[1].cpu().long(), got[2].cpu().long(), 1.0)
    lv.mean().backward()
    assert bool(torch.isfinite(lv).all()) and bool(torch.isfinite(x.grad).all())


class _FakeDataset:
    def __init__(self, value):
        self.value = value

    def read(self):
        return self.value


class _FakeGroup(dict):
    def keys(self):
        return list(dict.keys(self))


def test_batch_layout_is_split_like_the_reference_reader():
    """`BatchHDF5Reader._load_reads_batch` (mapped_signal_files.py:503-540): the reads of the real
    per-read fixture, packed the way `BatchHDF5Writer.write_curr_batch` (:593-645) packs them
    (concatenated arrays + `_lengths`, scalars as 1-d arrays, two batches), come back identical.
    (No genuine batch-layout file exists in the reference's test data; the dataset decoding under
    this is what the tests above validate.)"""
    info, reads = hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "mapped_remap_samref.hdf5"))
    assert len(reads) >= 3
    batches = _FakeGroup()
    for b, part in enumerate((reads[:2], reads[2:])):
        g = _FakeGroup()
        for key in ("Dacs", "Ref_to_signal", "Reference"):
            g[key] = _FakeDataset(np.concatenate([r[key] for r in part]))
            g[key + "_lengths"] = _FakeDataset(np.array([len(r[key]) for r in part], dtype=np.int32))
        for key in hdf5_lite.BATCH_SCALARS:
            g[key] = _FakeDataset(np.array([r[key] for r in part], dtype=np.float64))
        g["read_id"] = _FakeDataset([r["read_id"] for r in part])
        batches["Batch_%d" % b] = g
    back = hdf5_lite.reads_of_batches(batches)
    assert [r["read_id"] for r in back] == [r["read_id"] for r in reads]
    for a, b in zip(back, reads):
        for key, dt in hdf5_lite.BATCH_ARRAYS:
            assert a[key].dtype == dt and np.array_equal(a[key], b[key])
        assert all(a[k] == b[k] for k in hdf5_lite.BATCH_SCALARS)
    assert len(hdf5_lite.reads_of_batches(batches, limit=1)) == 1
    # inconsistent lengths are an error, not a silent truncation
    batches["Batch_0"]["Dacs_lengths"] = _FakeDataset(np.array([1, 2], dtype=np.int32))
    with pytest.raises(hdf5_lite.Hdf5Error, match="add up"):
        hdf5_lite.reads_of_batches(batches)


def test_variable_length_string_elements_decode_through_the_global_heap():
    """The 16-byte (length, collection address, object index) elements of an h5py
    `special_dtype(vlen=str)` dataset (mapped_signal_files.py:21, the read ids of a batch)."""
    import struct

    class FakeFile:
        heap = {(4096, 1): b"read-one", (4096, 2): b"r2\0\0\0\0", (8192, 7): b""}

        def _global_heap_object(self, addr, index):
            return self.heap[(addr, index)]

    dt = hdf5_lite._Datatype(9, 16, None, vlen_string=True, base=None)
    raw = np.frombuffer(struct.pack("<IQI", 8, 4096, 1) + struct.pack("<IQI", 2, 4096, 2) +
                        struct.pack("<IQI", 0, 0, 0), dtype="V16")
    assert hdf5_lite.decode_vlen(FakeFile(), dt, raw) == ["read-one", "r2", ""]


REFDATA = "/root/reference/test/data"


@pytest.mark.skipif(not os.path.isdir(REFDATA), reason="the reference's test data is only in the build container")
def test_every_hdf5_file_of_the_reference_test_data_parses():
    """All nine HDF5 containers under the reference's test/data (three mapped-signal files, five
    single-read fast5, one multi-read fast5): structure invariants, and the multi-read fast5 holds
    exactly the signals of the single-read files (two writers, two layouts of the same reads)."""
    import glob
    for path in sorted(glob.glob(os.path.join(REFDATA, "mapped_signal_file", "*.hdf5"))):
        info, reads = hdf5_lite.read_mapped_signal_file(path)
        assert info["version"] == 8 and len(reads) >= 2
        for r in reads:
            assert len(r["Ref_to_signal"]) == len(r["Reference"]) + 1 and np.all(np.diff(r["Ref_to_signal"]) >= 0)
            assert r["Ref_to_signal"][-1] <= len(r["Dacs"]) + 1
    multi = hdf5_lite.File(glob.glob(os.path.join(REFDATA, "multireads", "*.fast5"))[0])
    singles = {}
    for path in glob.glob(os.path.join(REFDATA, "reads", "*.fast5")):
        rg = hdf5_lite.File(path)["Raw/Reads"]
        (name,) = rg.keys()
        sig = rg[name]["Signal"].read()
        assert int(rg[name].attrs["duration"]) == len(sig)
        singles[str(rg[name].attrs["read_id"])] = sig
    assert len(singles) == 5
    seen = 0
    for key in multi.keys():
        raw = multi[key]["Raw"]
        rid = str(raw.attrs["read_id"])
        assert key == "read_" + rid
        if rid in singles:
            assert np.array_equal(raw["Signal"].read(), singles[rid])
            seen += 1
    assert seen == 5


def test_a_fractal_heap_that_saw_deletions_or_special_objects_is_refused_by_name(tmp_path):
    """The dense-group / dense-attribute walk reads a heap's managed objects back to back, which is what
    a write-once file looks like.  A heap whose own accounting does not add up to what the walk found
    (objects were deleted: holes, stale bytes), or that holds huge / tiny objects, must be refused with an
    error that says so and names the way out -- not mis-parsed.  Simulated on copies of the libhdf5-written
    1.8 fixture by editing the heap header's counters."""
    import struct
    src = open(os.path.join(HERE, "generated_v108.hdf5"), "rb").read()
    at = src.index(b"FRHP")
    # FRHP: sig 4, version 1, id length 2, filter length 2, flags 1, max managed size 4, then 8-byte fields:
    # next huge id, huge B-tree, FREE SPACE, free-space manager, managed, allocated, iterator, NOBJ,
    # huge size, HUGE COUNT, tiny size, TINY COUNT
    f0 = at + 14
    for what, off, match in (("free space", f0 + 16, "deleted"), ("huge count", f0 + 72, "huge"),
                             ("tiny count", f0 + 88, "tiny")):
        buf = bytearray(src)
        old, = struct.unpack_from("<Q", buf, off)
        struct.pack_into("<Q", buf, off, old + 3)
        path = tmp_path / ("tampered_%s.hdf5" % what.replace(" ", "_"))
        path.write_bytes(bytes(buf))
        with pytest.raises(hdf5_lite.Hdf5Error, match=match):
            f = hdf5_lite.File(str(path))
            hdf5_lite.read_mapped_signal_file(str(path))
            del f
    # the untouched file still reads
    assert hdf5_lite.read_mapped_signal_file(os.path.join(HERE, "generated_v108.hdf5"))[1]


@pytest.mark.parametrize("n", [35, 150, 2000, 40000])
def test_fractal_heap_with_a_partially_filled_root_indirect_block_reads(tmp_path, n):
    """Round-4 advisor finding: the heap accounting check compared the bytes found with `allocated - free -
    headers`, but libhdf5 books the free space of every direct block a root indirect block's rows CAN hold when
    the indirect block is created or doubled, before those blocks exist -- valid never-modified files of 35, 50,
    100, 2000 links were refused (the announced byte count even negative).  Fixtures written by libhdf5 1.10.6
    itself (gen_heap_fixtures.c): one group of n empty sub-groups (2, 7 and 22 direct blocks under one root indirect block).
    Round-5 advisor finding: none of them reaches the rows of CHILD indirect blocks (beyond the largest direct block, ~25 thousand
    links): 40000 links do -- a root indirect block with three children, 115 direct blocks."""
    import gzip
    path = os.path.join(HERE, "heap_%d.hdf5" % n)
    if not os.path.exists(path):
        path = str(tmp_path / ("heap_%d.hdf5" % n))
        with open(path, "wb") as fh:
            fh.write(gzip.open(os.path.join(HERE, "heap_%d.hdf5.gz" % n)).read())
    raw = open(path, "rb").read()
    assert raw.count(b"FRHP") >= 1 and raw.count(b"FHIB") == (4 if n == 40000 else 1)
    assert raw.count(b"FHDB") == {35: 2, 150: 7, 2000: 22, 40000: 115}[n]
    f = hdf5_lite.File(path)
    assert sorted(f["Reads"].keys()) == ["read_%05d" % r for r in range(n)]


def test_fractal_heap_with_child_indirect_blocks_that_lost_links_is_refused(tmp_path):
    """... and its every-11th-deleted variant (40000 links, child indirect blocks) is refused like the small one below."""
    import gzip
    path = str(tmp_path / "heap_40000_del.hdf5")
    with open(path, "wb") as fh:
        fh.write(gzip.open(os.path.join(HERE, "heap_40000_every11th_deleted.hdf5.gz")).read())
    with pytest.raises(hdf5_lite.Hdf5Error, match="deleted"):
        f = hdf5_lite.File(path)
        list(f["Reads"].keys())


def test_fractal_heap_that_really_lost_links_is_refused():
    """A file libhdf5 wrote and then unlinked every 7th sub-group from: the deleted links' bytes stay in the
    heap's blocks, equal-sized, so the byte accounting alone adds up for the FIRST `nobj` objects (the reader
    returned stale names); the walk now counts every parseable object and refuses the surplus."""
    with pytest.raises(hdf5_lite.Hdf5Error, match="deleted"):
        f = hdf5_lite.File(os.path.join(HERE, "heap_150_every7th_deleted.hdf5"))
        list(f["Reads"].keys())
