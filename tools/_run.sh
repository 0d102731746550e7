cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "viterbi" 2>&1 | tail -4
