#!/bin/bash
# same box: time the native step for several builds (GPE_EXTRA_CXXFLAGS variants), 1M and 16M particles
# usage: bash scripts/ab_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...   ("-" = no flags)
set -u
for v in "$@"; do
  flags="$v"; [ "$v" = "-" ] && flags=""
  GPE_EXTRA_CXXFLAGS="$flags" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "variant [$v]"
  for r in 1 2; do timeout -k 10 120 python scripts/time_step.py 1000000 200 2>&1 | grep "^n=" | cut -c1-150; done
  timeout -k 10 120 python scripts/time_step.py 16000000 40 2>&1 | grep "^n=" | cut -c1-150
done
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
