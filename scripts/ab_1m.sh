#!/bin/bash
# Same box, 1 M only: time the native step with several prebuilt libraries (scripts/build_variant.sh), interleaved.
# usage: bash scripts/ab_1m.sh <name> ...
L=gpu-physics-engine_amd/libgpe.so
cp $L /tmp/libgpe_default.so
for r in 1 2 3; do
  for v in "$@"; do
    cp gpurun_tmp/variants/$v.so $L || { echo "no variant $v"; continue; }
    echo -n "[$v] "; timeout -k 10 120 python scripts/time_step.py 1000000 400 2>&1 | grep "^n=" | cut -c1-200
  done
done
cp /tmp/libgpe_default.so $L
