cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crf or catmod or ragged or harness or fullsize or nonfinite or depend" > gpurun_out/r2/pytest_crf.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2/pytest_crf.log
tail -4 gpurun_out/r2/pytest_crf.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2/prof_crf5 -o crf -- python $GRAFT_REPO_ROOT/tools/crfbench.py --fwd --shapes short1,cfg2r,cfg5r,rowK --modes band1,band2,band4 --reps 10 > $GRAFT_REPO_ROOT/gpurun_out/r2/prof_crf5.log 2>&1
cd $GRAFT_REPO_ROOT; grep "band" gpurun_out/r2/prof_crf5.log | grep "grad" | tail -20
python tools/prof_by_shape.py gpurun_out/r2/prof_crf5/crf_results.db | grep band
