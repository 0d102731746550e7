cd /root/repo
export TK_BENCH_SHARE_GPU=1 TK_BENCH_STACKS_AFTER=400
timeout 500 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/two.out 2> gpurun_out/two.err
echo "rc=$?"
grep "probe\|capture\|rank\|Timeout" gpurun_out/two.err | grep -v "arena rank" | head -10
tail -1 gpurun_out/two.out | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value','n_gpus','ms_per_step')}, d['rccl'], d['roofline']['frac'], list(d.keys()))"
unset TK_BENCH_STACKS_AFTER
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_ranks or live_rccl" 2>&1 | tail -3
