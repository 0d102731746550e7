cd /root/repo
timeout 300 python tools/graph_op_probe.py 2>&1 | grep -v amdgpu
