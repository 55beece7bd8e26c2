#!/bin/bash
# same box: clustered / piled scenes (where the over-capacity launch carries the work), current vs reference kernel
set -u
run() {
  timeout -k 10 200 python scripts/time_evolution.py 1000000 3 3000 2>&1 | grep -v amdgpu | tail -2 | cut -c1-260
  timeout -k 10 300 python scripts/time_evolution.py 100000000 5 120 on 2>&1 | grep -v amdgpu | tail -3 | cut -c1-260
}
echo "current"; python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1; run
cp gpu-physics-engine_amd/csrc/k_native.hip /tmp/k_native_new.hip
cp gpu-physics-engine_amd/csrc/gpe_internal.h /tmp/gpe_internal_new.h
cp gpurun_tmp/k_native_head.hip.txt gpu-physics-engine_amd/csrc/k_native.hip
cp gpurun_tmp/gpe_internal_head.h.txt gpu-physics-engine_amd/csrc/gpe_internal.h
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
echo "reference commit"; run
cp /tmp/k_native_new.hip gpu-physics-engine_amd/csrc/k_native.hip
cp /tmp/gpe_internal_new.h gpu-physics-engine_amd/csrc/gpe_internal.h
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
