cd /root/repo
timeout 1200 python -m tests.helpers.fuzz_shapes --cases 14 --seed 77 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crf or band or catmod or fused or fuzz_shapes or fullsize or ragged or poison or write" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_post -o p -- python /root/repo/tools/crfbench.py --reps 20 --shapes cfg2r,cfg4 --modes band 2>&1 | grep "^band"
python /root/repo/tools/prof_by_shape.py /root/repo/gpurun_out/prof_post/p_results.db 2>/dev/null | grep -i "posterior" | head
cd /root/repo; timeout 600 python tools/pmc_traffic.py --ops crf:800:128:4000 2>&1 | grep -v amdgpu | head -3
