#!/bin/bash
# diagnostic: cost of the collide phases by omission (results are wrong in these builds!), then restore the product build
set -u
for skip in ${SKIPS:-0 1 3 7}; do
  GPE_EXTRA_CXXFLAGS="-DGPE_DBG_SKIP=$skip" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || exit 1
  echo "skip=$skip"; timeout -k 10 120 python scripts/time_step.py 16000000 40 2>&1 | tail -1 | cut -c1-120
done
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
