#!/bin/bash
# Same box: time the native step with several prebuilt libraries (scripts/build_variant.sh), swapping each in as
# libgpe.so.  usage: bash scripts/ab_libs.sh [-q] <name> ...     (-q: skip the 100 M leg)
set -u
quick=0; [ "$1" = "-q" ] && { quick=1; shift; }
L=gpu-physics-engine_amd/libgpe.so
cp $L /tmp/libgpe_default.so
for v in "$@"; do
  cp gpurun_tmp/variants/$v.so $L || { echo "no variant $v"; continue; }
  echo "variant [$v]"
  for r in 1 2; do timeout -k 10 120 python scripts/time_step.py 1000000 300 2>&1 | grep "^n=" | cut -c1-200; done
  timeout -k 10 120 python scripts/time_step.py 16000000 40 2>&1 | grep "^n=" | cut -c1-200
  [ $quick -eq 0 ] && timeout -k 10 200 python scripts/time_step.py 100000000 30 on 2>&1 | grep "^n=" | cut -c1-200
done
cp /tmp/libgpe_default.so $L
