#!/bin/bash
# same box: ms/step as the 1M scene relaxes (clusters form), current k_native.hip vs the reference commit's
set -u
echo "current"; python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
timeout -k 10 200 python scripts/time_evolution.py 1000000 3 3000 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
cp gpu-physics-engine_amd/csrc/k_native.hip /tmp/k_native_new.hip
cp gpu-physics-engine_amd/csrc/gpe_internal.h /tmp/gpe_internal_new.h
cp gpurun_tmp/k_native_head.hip.txt gpu-physics-engine_amd/csrc/k_native.hip
cp gpurun_tmp/gpe_internal_head.h.txt gpu-physics-engine_amd/csrc/gpe_internal.h
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
echo "reference commit"
timeout -k 10 200 python scripts/time_evolution.py 1000000 3 3000 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
cp /tmp/k_native_new.hip gpu-physics-engine_amd/csrc/k_native.hip
cp /tmp/gpe_internal_new.h gpu-physics-engine_amd/csrc/gpe_internal.h
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
