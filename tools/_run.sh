cd /root/repo
mkdir -p gpurun_out/ev4
timeout 900 python bench.py > gpurun_out/ev4/bench_default.json 2> gpurun_out/ev4/bench_default.err
timeout 900 python tools/pmc_traffic.py --ops logz:4000:256:0,logz:800:128:0,crf:800:128:4000,crf:4000:256:0,catmod:800:128:4000 --save gpurun_out/ev4/r2b > gpurun_out/ev4/pmc.log 2>&1
cut -c1-200 gpurun_out/ev4/bench_default.json; grep "x algorithmic" gpurun_out/ev4/pmc.log | head -3
