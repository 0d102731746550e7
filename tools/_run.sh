cd /root/repo
timeout 900 python -m pytest tests/test_beamsearch.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/_bm.py 2>&1 | grep -v amdgpu
