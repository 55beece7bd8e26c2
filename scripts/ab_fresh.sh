#!/bin/bash
# The fresh 1 M cloud (scripts/time_step.py 1000000 400) for several prebuilt libraries, twice each, interleaved, same box.
# usage: bash scripts/ab_fresh.sh <tag> <name> ...
set -u
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
L=$ROOT/gpu-physics-engine_amd/libgpe.so; cp $L /tmp/libgpe_default.so
cd $ROOT; : > $OUT/ab_fresh_$tag.txt
for r in 1 2; do
  for v in "$@"; do
    cp gpurun_tmp/variants/$v.so $L || { echo "no variant $v"; continue; }
    echo -n "[$v] " | tee -a $OUT/ab_fresh_$tag.txt; timeout -k 10 120 python scripts/time_step.py 1000000 400 2>&1 | grep "^n=" | cut -c1-200 | tee -a $OUT/ab_fresh_$tag.txt
  done
done
cp /tmp/libgpe_default.so $L
