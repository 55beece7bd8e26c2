#!/bin/bash
# diagnostic build with per-phase cycle stamps in the collide tiles, then restore the product build
set -u
GPE_EXTRA_CXXFLAGS="-DGPE_TILE_STAMPS" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || exit 1
for n in ${@:-1000000 16000000}; do timeout -k 10 120 python scripts/time_step.py $n 60 2>&1 | grep -v amdgpu.ids | tail -3; done
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
