#!/bin/bash
# Same box: time the native step at 1 M and 16 M with several prebuilt libraries (scripts/build_variant.sh).
L=gpu-physics-engine_amd/libgpe.so
cp $L /tmp/libgpe_default.so
for v in "$@"; do
  cp gpurun_tmp/variants/$v.so $L || { echo "no variant $v"; continue; }
  echo -n "[$v] "; timeout -k 10 120 python scripts/time_step.py 1000000 400 2>&1 | grep "^n=" | cut -c1-230
  echo -n "[$v] "; timeout -k 10 120 python scripts/time_step.py 16000000 60 2>&1 | grep "^n=" | cut -c1-230
done
cp /tmp/libgpe_default.so $L
