cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2/pytest_full.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2/pytest_full.log
tail -6 gpurun_out/r2/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --data store --no-rowk --no-pmc --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
