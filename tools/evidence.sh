#!/bin/bash
# The round's evidence files on one GPU box (what profiles/README.md's table cites), written under gpurun_out/$TAG/:
#   gpurun --timeout 2400 -- 'TAG=r6_v1 bash tools/evidence.sh'
# PMC passes never share a run with a trace domain other than --kernel-trace (MI355X guide; tools/pmc_traffic.py,
# tools/sq_counters.py wrap rocprofv3 themselves).  Every step is bounded by `timeout`.
cd "$(dirname "$0")/.."; TAG=${TAG:-r6}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
H=$(python -c "import bench; print(bench.kernel_hash())" 2>/dev/null | tail -1); echo "kernel hash $H" | tee $O/hash.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/show_bench.py $O/bench_default.json
for c in 4 5; do timeout 600 python bench.py --config $c --no-forced-group --no-varlen > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; done
# fuzz sweeps beside the profiler passes (CPU-heavy -- the float64 witness -- so NOT beside the bench lines above: the
# train step is bound by its host thread, two fuzz processes next to it halve its rate)
for s in ${FUZZ_SEEDS:-21 22}; do timeout 2000 python -m tests.helpers.fuzz_shapes --cases ${FUZZ_CASES:-100} --seed $s > $O/fuzz_shapes_$s.log 2>&1 & done
# kernel summary of the bench command under rocprofv3 (kernel trace only)
rm -rf /tmp/tk_prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/tk_prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-forced-group --no-cpu-baseline --no-pmc --no-varlen > /dev/null 2> $OLDPWD/$O/rocprof_bench.err)
DB=$(find /tmp/tk_prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB 30 > $O/bench_kernel_stats.txt; python tools/prof_by_shape.py $DB "%tk%" > $O/bench_loss_kernels_by_shape.txt; fi
find /tmp/tk_prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/bench_rocprofv3_kernel_stats.csv
# HBM traffic of the loss-path operators, shader-sequencer counters of kernel A
timeout 900 python tools/pmc_traffic.py --ops logz:4000:256:0,logz:800:128:0,crf:800:128:4000,crf:4000:256:0,catmod:800:128:4000 --save $O/${TAG%%_*} > $O/pmc.log 2>&1
timeout 900 python tools/sq_counters.py --save $O/${TAG%%_*}_sq_counters.json > $O/sq_counters.txt 2>&1
# operator timings outside the step
timeout 300 python tools/vitbench.py > $O/vitbench.txt 2>&1
for v in "" "--separate"; do echo "== crfops $v"; timeout 300 python tools/crfops.py --rowk --catmod --cfg5 $v 2>&1 | tail -1; done > $O/crfops.txt
timeout 300 python tools/crf_gate_probe.py --shapes tiny,t19,t37,t200,cfg2,cfg2r,narrow,pathbuf,lenramp,initramp,conframp,sharp,sharp13,sharp15,sharp17,sharp2,sharp3,sharp4,sharp2K,cfg4,cfg4r,cfg4rharsh,cfg4sharp2,cfg4sharp25,cfg4sharp3,cfg5,rowK,conf,confburst,confK,realnet,realnet_s2,realnet_s3,realfast,realfast_s2,realfast_s3,realnet_cm,realnet_cm_s13,realnet_cm_s2,realnet_cm_s25,realnet_cm_s3,realfast_cm,realfast_cm_s13,realfast_cm_s2,realfast_cm_s25,realfast_cm_s3 > $O/crf_gate_probe.log 2>&1
timeout 600 python tools/crf_gate_band_probe.py > $O/gate_by_band_width.txt 2>&1
wait
tail -n 2 $O/fuzz_shapes_*.log; grep -h FAIL $O/fuzz_shapes_*.log | head
ls -la $O
