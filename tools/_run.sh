cd /root/repo
timeout 1500 python -m tests.helpers.fuzz_shapes --cases 150 --seed 202 2>&1 | grep -n "FAIL\|fuzz:"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fuzz" 2>&1 | tail -2
