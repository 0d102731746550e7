#!/bin/bash
# Long fuzz sweeps on a GPU box, side by side with the -m gpu suite (round-3 evidence: profiles/r3_fuzz_summary.txt):
# four seeds of tests.helpers.fuzz_shapes x 140 cases, two seeds each of fuzz_beam and fuzz_prep x 100.
#   gpurun --timeout 3000 -- 'bash tools/fuzz_all.sh'
cd "$(dirname "$0")/.."; O=gpurun_out/fuzz; mkdir -p $O
# FUZZ_SEEDS="5 6 7 8" FUZZ_SEEDS2="3 4" bash tools/fuzz_all.sh   for other seeds
for s in ${FUZZ_SEEDS:-1 2 3 4}; do python -m tests.helpers.fuzz_shapes --cases 140 --seed $s > $O/shapes_$s.log 2>&1 & done
python -m pytest tests -m gpu -x -q > gpurun_out/fuzz_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/fuzz_pytest.log | tail -2
for s in ${FUZZ_SEEDS2:-1 2}; do python -m tests.helpers.fuzz_beam --cases 100 --seed $s > $O/beam_$s.log 2>&1 & done
for s in ${FUZZ_SEEDS2:-1 2}; do python -m tests.helpers.fuzz_prep --cases 100 --seed $s > $O/prep_$s.log 2>&1 & done
wait
tail -n 1 $O/*.log; grep -h FAIL $O/*.log | head -20
