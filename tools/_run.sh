cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hdf5_reader.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --data store --mapped-signal tests/golden/mapped_signal/mapped_reads_0.hdf5 --no-rowk --no-pmc --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
