cd /root/repo; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crf or band or catmod or fused or poison or fuzz_shapes or fullsize or ragged" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_g -o p -- python /root/repo/tools/crfbench.py --reps 20 --shapes cfg2r,cfg4,cfg5r,rowK --modes band 2>&1 | grep "^band"
python /root/repo/tools/prof_by_shape.py /root/repo/gpurun_out/prof_g/p_results.db 2>/dev/null | grep -i "sweep"
