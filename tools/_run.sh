cd /root/repo
GPU_MAX_HW_QUEUES=4 timeout 300 python tools/overlap_probe.py 2>&1 | grep -v amdgpu
