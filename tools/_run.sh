cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2/final1; mkdir -p $O
python bench.py > $O/r2_bench_v1_default.json 2> $O/bench.err; tail -c 400 $O/r2_bench_v1_default.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/benchprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $O/benchprof/bench_results.db 40 > $O/r2_bench_v1_kernel_stats.txt; head -30 $O/r2_bench_v1_kernel_stats.txt
python tools/prof_by_shape.py $O/benchprof/bench_results.db | grep -i "logz\|crf\|indices\|seqoff" > $O/r2_bench_v1_loss_kernels_by_shape.txt; cat $O/r2_bench_v1_loss_kernels_by_shape.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/crfprof -o crf -- python $GRAFT_REPO_ROOT/tools/crfbench.py --fwd --shapes cfg2,cfg2r,cfg4,cfg5r,rowK --modes band,ckpt --reps 10 > $GRAFT_REPO_ROOT/$O/r2_crfbench_v6.log 2>/dev/null
cd $GRAFT_REPO_ROOT; grep -v amdgpu $O/r2_crfbench_v6.log | grep band
python tools/prof_by_shape.py $O/crfprof/crf_results.db | grep "crf\|indices\|seqoff" > $O/r2_crf_v6_by_shape.txt
timeout 600 python tools/pmc_traffic.py --ops logz:4000:256:0,crf:4000:256:0,logz:800:128:0,crf:800:128:4000,catmod:800:128:4000 --save $O/r2 > $O/r2_pmc_v2.log 2>&1; grep -v "^   " $O/r2_pmc_v2.log
./tools/lab_latlab > $O/r2_latlab.txt 2>&1
rm -rf $O/benchprof $O/crfprof
