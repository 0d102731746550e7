cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or logz" 2>&1 | tail -2
timeout 600 python bench.py --no-rowk --no-pmc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['loss_path']['gpu_ms'])"
