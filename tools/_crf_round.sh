#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd /root/repo
rm -rf gpurun_out/prof_crf
rocprofv3 --kernel-trace -d gpurun_out/prof_crf -o crf -- python tools/crfbench.py --shapes cfg2r,cfg5,rowK --modes band --reps 5 > /dev/null 2>&1
python tools/prof_by_shape.py gpurun_out/prof_crf/crf_results.db "%crf_band_sweep%"
