#!/bin/bash
# A/B lab builds: tools/build_variant.sh NAME [-DFLAG ...] -> tools/lab/lib_NAME.so (the lab library with crf_band.hip
# recompiled under the extra flags; use with TAIYAKI_AMD_LAB_LIB=tools/lab/lib_NAME.so)
set -e
cd "$(dirname "$0")/../taiyaki_amd/csrc"
NAME=$1; shift
make -s -j8 libtaiyaki_amd_flipflop_lab.so
OUT=../../tools/lab
FILES=${VARIANT_FILES:-crf_band.hip}
OBJS=""
for f in $FILES; do
  EXTRA=""; [ "$f" = crf_band.hip ] && [ -z "$VARIANT_SLP" ] && EXTRA="-fno-slp-vectorize"     # (the Makefile's per-file flag)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DTK_LAB $EXTRA "$@" -c -o $OUT/${NAME}_${f%.hip}.o $f
  OBJS="$OBJS $OUT/${NAME}_${f%.hip}.o"
done
REST=""
for f in c_api logz_kernels crf_kernels crf_band viterbi_kernels; do
  case " $FILES " in *" $f.hip "*) ;; *) REST="$REST lab_$f.o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o $OUT/lib_$NAME.so $OBJS $REST qscore_kernels.o clip_kernels.o chunk_kernels.o remap_kernels.o beam_kernels.o
echo built $OUT/lib_$NAME.so
