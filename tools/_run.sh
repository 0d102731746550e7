cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2/pytest_full.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2/pytest_full.log
tail -4 gpurun_out/r2/pytest_full.log
timeout 600 python tools/crfbench.py --shapes cfg2r,rowK --modes band --reps 10 2>&1 | grep -v amdgpu
