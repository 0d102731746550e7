/* gen_hdf5_fixtures.c -- writes two small mapped-signal files with the HDF5 LIBRARY itself (the
 * build container has libhdf5 1.10.6 under /opt/conda but no h5py, so the reference's Python writers
 * cannot run here): fixtures that pin taiyaki_amd/hdf5_lite.py's decoding of what those writers
 * make the library produce.  Test infrastructure; see make_v108_fixtures.sh.
 *
 *   per-read file, HDF5 1.8 layout -- what PerReadHDF5Writer's
 *       h5py.File(filename, 'w', libver='v108', track_order=True)     (mapped_signal_files.py:372)
 *       g.create_dataset(k, data=v, compression='gzip', shuffle=True) (:401), g.attrs[k] = v (:403)
 *     ask for: version-2 object headers, creation order tracked and indexed, link messages (compact)
 *     and fractal heaps (dense: NREADS groups under Reads/), chunked + shuffle + deflate datasets,
 *     variable-length UTF-8 string attributes; every other read carries ten attributes (> 8: dense
 *     attribute storage), and the root gets the `read_ids` variable-length string dataset of :383-391.
 *   batch file, classic layout -- what BatchHDF5Writer's h5py.File(filename, 'w') (:582) and
 *     write_curr_batch (:593-650) produce: Batches/Batch_k with the per-read arrays concatenated,
 *     `<key>_lengths` (int32), float64 columns for the scalars, a variable-length string column
 *     `read_id`, everything gzip-compressed (the numeric ones shuffled), and `read_ids` in the root.
 *
 * All values follow formulas the test re-computes (tests/test_hdf5_reader.py):
 *   read r (0-based):  nsig = 40 + 3 r, nref = 5 + r,
 *   Dacs[i] = (i * 7 + r * 13) % 1000 - 300,  Reference[i] = (i + r) % 4,
 *   Ref_to_signal[i] = (i * nsig) / nref  (integer division; i = 0 .. nref),
 *   shift = 1.5 + r, scale = 0.25 * (r + 1), range = 1400 + r, offset = 10 - r, digitisation = 8192,
 *   read id = "%08x-aaaa-4bbb-8ccc-%012x" % (r * 2654435761 mod 2^32, r).
 */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { if ((x) < 0) { fprintf(stderr, "HDF5 call failed at line %d\n", __LINE__); exit(1); } } while (0)

static void read_id(int r, char *out) {
    sprintf(out, "%08x-aaaa-4bbb-8ccc-%012x", (unsigned)((unsigned long long)r * 2654435761ull & 0xffffffffull), r);
}
static int nsig_of(int r) { return 40 + 3 * r; }
static int nref_of(int r) { return 5 + r; }

static void str_attr(hid_t obj, const char *name, const char *val) {
    hid_t t = H5Tcopy(H5T_C_S1), s = H5Screate(H5S_SCALAR);
    CHECK(H5Tset_size(t, H5T_VARIABLE));
    CHECK(H5Tset_cset(t, H5T_CSET_UTF8));
    hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(a);
    CHECK(H5Awrite(a, t, &val));
    H5Aclose(a); H5Sclose(s); H5Tclose(t);
}
static void f64_attr(hid_t obj, const char *name, double v) {
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, H5T_IEEE_F64LE, s, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(a);
    CHECK(H5Awrite(a, H5T_NATIVE_DOUBLE, &v));
    H5Aclose(a); H5Sclose(s);
}
static void i64_attr(hid_t obj, const char *name, long long v) {
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, H5T_STD_I64LE, s, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(a);
    CHECK(H5Awrite(a, H5T_NATIVE_LLONG, &v));
    H5Aclose(a); H5Sclose(s);
}
/* h5py's create_dataset(data=v, compression='gzip', shuffle=True): one chunk shape guess; here the
 * whole array in chunks of `chunk` elements, shuffle + deflate level 4 */
static void dataset(hid_t g, const char *name, hid_t filetype, hid_t memtype, const void *data, hsize_t n,
                    hsize_t chunk, int shuffle) {
    hid_t s = H5Screate_simple(1, &n, NULL), p = H5Pcreate(H5P_DATASET_CREATE);
    if (chunk > n) chunk = n;
    CHECK(H5Pset_chunk(p, 1, &chunk));
    if (shuffle) CHECK(H5Pset_shuffle(p));
    CHECK(H5Pset_deflate(p, 4));
    hid_t d = H5Dcreate2(g, name, filetype, s, H5P_DEFAULT, p, H5P_DEFAULT);
    CHECK(d);
    CHECK(H5Dwrite(d, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, data));
    H5Dclose(d); H5Pclose(p); H5Sclose(s);
}
static void vlen_str_dataset(hid_t g, const char *name, char **vals, hsize_t n) {
    hid_t t = H5Tcopy(H5T_C_S1);
    CHECK(H5Tset_size(t, H5T_VARIABLE));
    CHECK(H5Tset_cset(t, H5T_CSET_UTF8));
    hid_t s = H5Screate_simple(1, &n, NULL), p = H5Pcreate(H5P_DATASET_CREATE);
    hsize_t chunk = n;
    CHECK(H5Pset_chunk(p, 1, &chunk));
    CHECK(H5Pset_deflate(p, 4));
    hid_t d = H5Dcreate2(g, name, t, s, H5P_DEFAULT, p, H5P_DEFAULT);
    CHECK(d);
    CHECK(H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, vals));
    H5Dclose(d); H5Pclose(p); H5Sclose(s); H5Tclose(t);
}
static void root_attrs(hid_t f) {
    i64_attr(f, "version", 8);
    str_attr(f, "alphabet", "ACGT");
    str_attr(f, "collapse_alphabet", "ACGT");
    str_attr(f, "mod_long_names", "");
}
static void fill(int r, int16_t *dacs, int32_t *rts, int16_t *ref) {
    const int ns = nsig_of(r), nr = nref_of(r);
    for (int i = 0; i < ns; ++i) dacs[i] = (int16_t)((i * 7 + r * 13) % 1000 - 300);
    for (int i = 0; i < nr; ++i) ref[i] = (int16_t)((i + r) % 4);
    for (int i = 0; i <= nr; ++i) rts[i] = (int32_t)(((long long)i * ns) / nr);
}

static void per_read_file(const char *path, int nreads) {
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS), fcpl = H5Pcreate(H5P_FILE_CREATE);
    CHECK(H5Pset_libver_bounds(fapl, H5F_LIBVER_V18, H5F_LIBVER_LATEST));         /* h5py libver='v108' */
    CHECK(H5Pset_link_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED));   /* track_order=True */
    CHECK(H5Pset_attr_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED));
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, fcpl, fapl);
    CHECK(f);
    root_attrs(f);
    hid_t gcpl = H5Pcreate(H5P_GROUP_CREATE);
    CHECK(H5Pset_link_creation_order(gcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED));
    CHECK(H5Pset_attr_creation_order(gcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED));
    hid_t reads = H5Gcreate2(f, "Reads", H5P_DEFAULT, gcpl, H5P_DEFAULT);
    CHECK(reads);
    char **ids = malloc(sizeof(char *) * nreads);
    for (int r = 0; r < nreads; ++r) {
        ids[r] = malloc(64);
        read_id(r, ids[r]);
        hid_t g = H5Gcreate2(reads, ids[r], H5P_DEFAULT, gcpl, H5P_DEFAULT);
        CHECK(g);
        int16_t dacs[4096], ref[2048];
        int32_t rts[2049];
        fill(r, dacs, rts, ref);
        dataset(g, "Dacs", H5T_STD_I16LE, H5T_NATIVE_INT16, dacs, nsig_of(r), 64, 1);
        dataset(g, "Ref_to_signal", H5T_STD_I32LE, H5T_NATIVE_INT32, rts, nref_of(r) + 1, 1024, 1);
        dataset(g, "Reference", H5T_STD_I16LE, H5T_NATIVE_INT16, ref, nref_of(r), 1024, 1);
        f64_attr(g, "shift_frompA", 1.5 + r);
        f64_attr(g, "scale_frompA", 0.25 * (r + 1));
        f64_attr(g, "range", 1400.0 + r);
        f64_attr(g, "offset", 10.0 - r);
        f64_attr(g, "digitisation", 8192.0);
        str_attr(g, "read_id", ids[r]);
        if (r % 2 == 0) {               /* ten attributes: beyond the compact limit of 8 */
            f64_attr(g, "mapping_score", 100.0 + 0.5 * r);
            str_attr(g, "mapping_method", "generated");
            f64_attr(g, "extra_a", (double)r);
            f64_attr(g, "extra_b", -(double)r);
        }
        H5Gclose(g);
    }
    vlen_str_dataset(f, "read_ids", ids, nreads);
    H5Gclose(reads); H5Pclose(gcpl);
    CHECK(H5Fclose(f));
    H5Pclose(fapl); H5Pclose(fcpl);
}

static void batch_file(const char *path, int nreads, int batch_size) {
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);          /* h5py.File(filename, 'w') */
    CHECK(f);
    root_attrs(f);
    hid_t batches = H5Gcreate2(f, "Batches", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(batches);
    char **ids = malloc(sizeof(char *) * nreads);
    for (int r = 0; r < nreads; ++r) {
        ids[r] = malloc(64);
        read_id(r, ids[r]);
    }
    for (int b = 0, r0 = 0; r0 < nreads; ++b, r0 += batch_size) {
        const int n = (nreads - r0 < batch_size) ? nreads - r0 : batch_size;
        char name[32];
        sprintf(name, "Batch_%d", b);
        hid_t g = H5Gcreate2(batches, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        CHECK(g);
        int16_t *dacs = malloc(2 * 4096 * n), *ref = malloc(2 * 2048 * n);
        int32_t *rts = malloc(4 * 2049 * n), *ld = malloc(4 * n), *lr = malloc(4 * n), *lt = malloc(4 * n);
        double *col[5];
        for (int k = 0; k < 5; ++k) col[k] = malloc(8 * n);
        size_t od = 0, orf = 0, ot = 0;
        for (int i = 0; i < n; ++i) {
            const int r = r0 + i;
            fill(r, dacs + od, rts + ot, ref + orf);
            ld[i] = nsig_of(r); lr[i] = nref_of(r); lt[i] = nref_of(r) + 1;
            od += ld[i]; orf += lr[i]; ot += lt[i];
            col[0][i] = 1.5 + r; col[1][i] = 0.25 * (r + 1); col[2][i] = 1400.0 + r; col[3][i] = 10.0 - r;
            col[4][i] = 8192.0;
        }
        dataset(g, "Dacs", H5T_STD_I16LE, H5T_NATIVE_INT16, dacs, od, 1024, 1);
        dataset(g, "Dacs_lengths", H5T_STD_I32LE, H5T_NATIVE_INT32, ld, n, 1024, 1);
        dataset(g, "Ref_to_signal", H5T_STD_I32LE, H5T_NATIVE_INT32, rts, ot, 1024, 1);
        dataset(g, "Ref_to_signal_lengths", H5T_STD_I32LE, H5T_NATIVE_INT32, lt, n, 1024, 1);
        dataset(g, "Reference", H5T_STD_I16LE, H5T_NATIVE_INT16, ref, orf, 1024, 1);
        dataset(g, "Reference_lengths", H5T_STD_I32LE, H5T_NATIVE_INT32, lr, n, 1024, 1);
        const char *cn[5] = {"shift_frompA", "scale_frompA", "range", "offset", "digitisation"};
        for (int k = 0; k < 5; ++k) dataset(g, cn[k], H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, col[k], n, 1024, 1);
        vlen_str_dataset(g, "read_id", ids + r0, n);
        H5Gclose(g);
    }
    vlen_str_dataset(f, "read_ids", ids, nreads);
    H5Gclose(batches);
    CHECK(H5Fclose(f));
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s per_read_v108.hdf5 batch.hdf5\n", argv[0]);
        return 2;
    }
    per_read_file(argv[1], 60);
    batch_file(argv[2], 23, 10);
    return 0;
}
