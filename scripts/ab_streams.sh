#!/bin/bash
# same box: one stream vs two streams for the over-capacity tiles, fresh and relaxed scenes
for rep in 1 2; do
for m in 0 1; do
  echo "GPE_OVERFLOW_STREAM=$m"
  GPE_OVERFLOW_STREAM=$m timeout -k 10 200 python scripts/time_step.py 1000000 2000 2>&1 | grep -v amdgpu | tail -1 | cut -c1-260
done
done
for m in 0 1; do
  echo "GPE_OVERFLOW_STREAM=$m relaxed"
  GPE_OVERFLOW_STREAM=$m timeout -k 10 200 python scripts/time_evolution.py 1000000 3 3000 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300
done
