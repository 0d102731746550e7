cd /tmp && export TMPDIR=/tmp
for lds in 84000; do
  echo "== TK_CRF_SWEEP_LDS=$lds"
  TK_CRF_SWEEP_LDS=$lds timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_lds$lds -o p -- python /root/repo/tools/crfbench.py --reps 20 --shapes cfg2r,cfg4,cfg5r --modes band 2>&1 | grep "^band\|rror"
  python /root/repo/tools/prof_by_shape.py /root/repo/gpurun_out/prof_lds$lds/p_results.db 2>/dev/null | grep -i "sweep"
done
