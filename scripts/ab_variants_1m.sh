#!/bin/bash
# same box: time the native step at 1M (three runs) for several builds (GPE_EXTRA_CXXFLAGS variants; "-" = no flags)
set -u
for v in "$@"; do
  flags="$v"; [ "$v" = "-" ] && flags=""
  GPE_EXTRA_CXXFLAGS="$flags" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "variant [$v]"
  for r in 1 2 3; do timeout -k 10 120 python scripts/time_step.py 1000000 300 2>&1 | grep "^n=" | cut -c1-170; done
done
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
