#!/bin/bash
# ms/step of the 100 M gravity-on scene at a few marks for several prebuilt libraries (one after the other, same box).
# usage: bash scripts/gpu_soak_marks.sh <tag> "<marks>" <name> ...
set -u
tag=$1; marks=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
L=$ROOT/gpu-physics-engine_amd/libgpe.so; cp $L /tmp/libgpe_default.so
cd $ROOT; : > $OUT/soak_marks_$tag.txt
for v in "$@"; do
  cp gpurun_tmp/variants/$v.so $L || { echo "no variant $v"; continue; }
  echo "[$v]" | tee -a $OUT/soak_marks_$tag.txt
  timeout -k 10 500 python scripts/soak_pairs.py 100000000 100 $marks 2>&1 | tee -a $OUT/soak_marks_$tag.txt
  rc=${PIPESTATUS[0]}; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo timeout; cp /tmp/libgpe_default.so $L; exit 1; fi
done
cp /tmp/libgpe_default.so $L
