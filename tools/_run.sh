cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_step or hybrid or bench" > gpurun_out/r2/pytest_b.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2/pytest_b.log
tail -5 gpurun_out/r2/pytest_b.log
timeout 600 python tools/pmc_traffic.py --ops logz:4000:256:0,crf:4000:256:0,logz:800:128:0,crf:800:128:4000 --save gpurun_out/r2/r2 > gpurun_out/r2/pmc_v1.log 2>&1; cat gpurun_out/r2/pmc_v1.log | tail -30
( time timeout 900 python bench.py ) > gpurun_out/r2/bench_v2.log 2> gpurun_out/r2/bench_v2.err; tail -1 gpurun_out/r2/bench_v2.log; tail -3 gpurun_out/r2/bench_v2.err
for c in 4 5 1; do timeout 900 python bench.py --config $c --no-rowk --no-pmc --no-cpu-baseline > gpurun_out/r2/bench_cfg$c.log 2> gpurun_out/r2/bench_cfg$c.err; tail -1 gpurun_out/r2/bench_cfg$c.log | cut -c1-600; tail -2 gpurun_out/r2/bench_cfg$c.err; done
