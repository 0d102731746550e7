cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sub_batches or hybrid or whole_step or catmod_model or two_ranks" 2>&1 | tail -6
