#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd /root/repo
python -m pytest tests/test_gpu_parity.py -q -k "disowns or sharpened or log_prob or band or catmod or fused" 2>&1 | tail -4
python tools/crf_gate_probe.py --shapes cfg4,cfg4free,cfg4w1 2>&1 | grep -v amdgpu.ids
python tools/crfbench.py --shapes cfg4 --modes band,ckpt 2>&1 | grep -v amdgpu.ids
python tools/pmc_traffic.py --ops crf:800:128:4000,crf:4000:256:0,catmod:800:128:4000 --save gpurun_out/r3 2>&1 | grep -v amdgpu.ids
