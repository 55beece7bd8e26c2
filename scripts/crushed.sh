#!/bin/bash
# the 100M gravity scene up to the crushed-pile regime, for two handover thresholds (diagnostic builds)
set -u
for h in 16384 4096; do
  GPE_EXTRA_CXXFLAGS="-DGPE_WINDOW_HANDOVER=$h" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || exit 1
  echo "handover=$h"
  timeout -k 10 500 python scripts/time_evolution.py 100000000 4 500 on 2>&1 | grep -v amdgpu | cut -c1-330
done
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
