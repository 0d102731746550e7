#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd /root/repo
python tools/crf_gate_probe.py 2>&1 | grep -v amdgpu.ids
python tools/crfbench.py --shapes cfg2r,cfg4,cfg5,rowK --modes band 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/prof_crf
rocprofv3 --kernel-trace -d gpurun_out/prof_crf -o crf -- python tools/crfbench.py --shapes cfg2r,cfg4,cfg5,rowK --modes band --reps 5 > /dev/null 2>&1
python tools/prof_by_shape.py gpurun_out/prof_crf/crf_results.db "%crf_band%"
python -m pytest tests/test_gpu_parity.py -q -k "crf or catmod or fused or fuzz or fullsize or mean_loss or reproducible" 2>&1 | tail -3
