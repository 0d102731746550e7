cd /root/repo
echo "== BK=8"; timeout 300 python tools/crfbench.py --reps 20 --shapes cfg2r,cfg5r,rowK,cfg4 --modes band 2>&1 | grep -v amdgpu.ids
echo "== BK=16"; TAIYAKI_AMD_LIB=/root/repo/tools/lab_bk16.so timeout 300 python tools/crfbench.py --reps 20 --shapes cfg2r,cfg5r,rowK,cfg4 --modes band 2>&1 | grep -v amdgpu.ids
TAIYAKI_AMD_LIB=/root/repo/tools/lab_bk16.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crf or band or catmod or fused or fuzz_shapes or fullsize or ragged or poison" 2>&1 | tail -4
