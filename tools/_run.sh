cd /root/repo
for seed in 101 202 303; do
  timeout 1500 python -m tests.helpers.fuzz_shapes --cases 150 --seed $seed 2>&1 | grep -v amdgpu.ids > gpurun_out/fuzz_shapes_$seed.log
  grep -n "FAIL\|fuzz:" gpurun_out/fuzz_shapes_$seed.log
done
timeout 1200 python -m tests.helpers.fuzz_prep --cases 200 --seed 404 2>&1 | grep -n "FAIL\|fuzz:"
timeout 1200 python -m tests.helpers.fuzz_beam --cases 200 --seed 505 2>&1 | grep -n "FAIL\|fuzz_beam:"
