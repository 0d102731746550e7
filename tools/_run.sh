cd /root/repo
mkdir -p gpurun_out/ev3
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/ev3/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/ev3/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/ev3/bench_default.json 2> gpurun_out/ev3/bench_default.err
timeout 900 python tools/pmc_traffic.py --ops logz:4000:256:0,logz:800:128:0,crf:800:128:4000,crf:4000:256:0,catmod:800:128:4000 --save gpurun_out/ev3/r2b > gpurun_out/ev3/pmc.log 2>&1
cat gpurun_out/ev3/pytest_gpu.txt; cut -c1-200 gpurun_out/ev3/bench_default.json; grep "x algorithmic" gpurun_out/ev3/pmc.log
