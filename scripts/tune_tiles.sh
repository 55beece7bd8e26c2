#!/bin/bash
# threads per tile x window capacity (tiles per CU follow from the LDS size): diagnostic builds on the box
set -u
for cfg in "512 1200" "384 880" "384 1200" "256 880"; do
  set -- $cfg
  GPE_EXTRA_CXXFLAGS="-DGPE_NAT_THREADS=$1 -DGPE_CAP_MAIN=$2" python gpu-physics-engine_amd/build.py --force > gpurun_out/tune_build.log 2>&1 || { echo "build failed: $cfg"; tail -3 gpurun_out/tune_build.log; continue; }
  echo "threads=$1 cap=$2"
  for n in 1000000 16000000; do timeout -k 10 200 python scripts/time_step.py $n 40 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-140; done
done
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
