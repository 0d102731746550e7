#!/bin/bash
# tools/ab_variants.sh "vbase vpair ..." : crfops + bitcmp per variant library (tools/lab/lib_NAME.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
first=""
for rep in 1 2; do
for v in $1; do
  L=$PWD/tools/lab/lib_$v.so
  echo "== $v (rep $rep): $(TAIYAKI_AMD_LIB=$L timeout 300 python tools/crfops.py --rowk --catmod --cfg5 --shapes d:800:256:4000,e:1200:128:6000::cm 2>&1 | tail -1)"
done; done
for v in $1; do
  L=$PWD/tools/lab/lib_$v.so
  if [ -z "$first" ]; then first=$v; TAIYAKI_AMD_LIB=$L timeout 300 python tools/crf_bitcmp.py /tmp/bit_$v.npz 2>&1 | tail -1;
  else echo "bitcmp $v vs $first: $(TAIYAKI_AMD_LIB=$L timeout 300 python tools/crf_bitcmp.py /tmp/bit_$v.npz /tmp/bit_$first.npz 2>&1 | tail -1)"; fi
done
