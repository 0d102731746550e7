#!/bin/bash
# one GPU round for kernel A: parity probe, timings, per-kernel split, per-phase stamps
cd /tmp && export TMPDIR=/tmp; cd /root/repo
python tools/crf_gate_probe.py 2>&1 | grep -v amdgpu.ids
python tools/crfbench.py --shapes cfg2r,cfg5,rowK --modes band,band2 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/prof_crf
rocprofv3 --kernel-trace -d gpurun_out/prof_crf -o crf -- python tools/crfbench.py --shapes cfg2r,cfg4,cfg5,rowK --modes band --reps 5 > /dev/null 2>&1
python tools/prof_by_shape.py gpurun_out/prof_crf/crf_results.db "%crf%"
TK_CRF_STAMPS=1 TAIYAKI_AMD_LIB=/root/repo/tools/lab_timing python tools/crfbench.py --shapes cfg2r --modes band --reps 1 2>&1 | grep phase | tail -4
