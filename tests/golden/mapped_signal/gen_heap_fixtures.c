/* gen_heap_fixtures.c -- files whose dense link storage (a fractal heap) has a PARTIALLY FILLED root
 * indirect block: one group "Reads" with N empty sub-groups "read_%05d", written with the newest
 * format bounds and nothing ever deleted.  The library adds the free space of every direct block of a
 * root indirect block's rows to the heap header's "free space in managed blocks" when the indirect block
 * is created or doubled -- BEFORE those blocks are allocated -- so the header's accounting is on the
 * managed space, not on the allocated space (round-4 advisor finding against taiyaki_amd/hdf5_lite.py).
 * BUILD CONTAINER ONLY (libhdf5 1.10.6 under /opt/conda); see make_v108_fixtures.sh.  Test infrastructure.
 *   usage: gen_heap_fixtures out.hdf5 N [K]     (K: then unlink every K-th sub-group again -- a heap that HAS
 *                                               seen deletions, which hdf5_lite must refuse by name)
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { if ((x) < 0) { fprintf(stderr, "HDF5 call failed at line %d\n", __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    if (argc != 3 && argc != 4) return 2;
    const int n = atoi(argv[2]), k = argc == 4 ? atoi(argv[3]) : 0;
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
    CHECK(H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST));
    hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
    CHECK(f);
    hid_t g = H5Gcreate2(f, "Reads", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(g);
    for (int r = 0; r < n; ++r) {
        char name[32];
        sprintf(name, "read_%05d", r);
        hid_t s = H5Gcreate2(g, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        CHECK(s);
        H5Gclose(s);
    }
    for (int r = k - 1; k > 0 && r < n; r += k) {
        char name[32];
        sprintf(name, "read_%05d", r);
        CHECK(H5Ldelete(g, name, H5P_DEFAULT));
    }
    H5Gclose(g);
    H5Fclose(f);
    H5Pclose(fapl);
    return 0;
}
