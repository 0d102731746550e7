#!/bin/bash
# Writes the HDF5-1.8-layout and batch-layout fixtures of tests/test_hdf5_reader.py.  BUILD CONTAINER
# ONLY: it needs the HDF5 library and its tools (here: /opt/conda, HDF5 1.10.6 -- there is no h5py, so
# the reference's Python writers cannot run) and, for the re-written reference files, /root/reference.
#
#   mapped_reads_0_v108.hdf5   the reference's own test file (test/data/mapped_signal_file/
#                              mapped_reads_0.hdf5, classic layout) re-written by `h5repack` with the
#                              bounds h5py's libver='v108' sets (low = V18): the SAME reads in the
#                              1.8 layout (superblock 2, version-2 object headers, link messages)
#   generated_v108.hdf5        60 reads written by the library with PerReadHDF5Writer's options
#   generated_batch.hdf5       23 reads in BatchHDF5Writer's layout, three batches
set -e
here=$(cd "$(dirname "$0")" && pwd)
export PATH=/opt/conda/bin:$PATH LD_LIBRARY_PATH=/opt/conda/lib
h5repack -j 1 -k 2 /root/reference/test/data/mapped_signal_file/mapped_reads_0.hdf5 "$here/mapped_reads_0_v108.hdf5"
gcc -O1 -I/opt/conda/include -o /tmp/gen_hdf5_fixtures "$here/gen_hdf5_fixtures.c" -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib
/tmp/gen_hdf5_fixtures "$here/generated_v108.hdf5" "$here/generated_batch.hdf5"
ls -la "$here"/*.hdf5
# fractal heaps with a partially filled root indirect block (round 5; gen_heap_fixtures.c): N sub-groups in
# one group, newest format bounds, nothing deleted -- and one file that did see deletions
gcc -O1 -I/opt/conda/include -o /tmp/gen_heap_fixtures "$here/gen_heap_fixtures.c" -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib
/tmp/gen_heap_fixtures "$here/heap_35.hdf5" 35
/tmp/gen_heap_fixtures "$here/heap_150.hdf5" 150
/tmp/gen_heap_fixtures "$here/heap_2000.hdf5" 2000 && gzip -9 -f "$here/heap_2000.hdf5"
/tmp/gen_heap_fixtures "$here/heap_150_every7th_deleted.hdf5" 150 7
# child indirect blocks (round 6): beyond ~25 thousand links the root indirect block's rows hold indirect blocks
/tmp/gen_heap_fixtures "$here/heap_40000.hdf5" 40000 && gzip -9 -f "$here/heap_40000.hdf5"
/tmp/gen_heap_fixtures "$here/heap_40000_every11th_deleted.hdf5" 40000 11 && gzip -9 -f "$here/heap_40000_every11th_deleted.hdf5"
