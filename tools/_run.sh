cd /root/repo
timeout 900 python -m pytest tests/test_beamsearch.py -x -q -m gpu 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_beam -o b -- python /root/repo/tools/beambench.py 2>&1 | grep "^beam"
python /root/repo/tools/rocpd_stats.py /root/repo/gpurun_out/prof_beam/b_results.db 2>/dev/null | head -8
