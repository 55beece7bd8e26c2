#!/bin/bash
# same box: time the native step at 1M (twice), 16M and 100M for several builds (GPE_EXTRA_CXXFLAGS variants; "-" = no flags)
set -u
for v in "$@"; do
  flags="$v"; [ "$v" = "-" ] && flags=""
  GPE_EXTRA_CXXFLAGS="$flags" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "variant [$v]"
  for r in 1 2; do timeout -k 10 120 python scripts/time_step.py 1000000 300 2>&1 | grep "^n=" | cut -c1-170; done
  timeout -k 10 120 python scripts/time_step.py 16000000 40 2>&1 | grep "^n=" | cut -c1-170
  timeout -k 10 200 python scripts/time_step.py 100000000 30 on 2>&1 | grep "^n=" | cut -c1-170
done
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
