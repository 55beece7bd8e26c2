#!/bin/bash
# same box: bench.py (1M) with the current k_native.hip vs the one of the last commit
set -u
run() { for r in 1 2 3; do timeout -k 10 120 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   ms/step %.4f  collide %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; done; }
echo "current"; python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1; run
cp gpu-physics-engine_amd/csrc/k_native.hip /tmp/k_native_new.hip
cp gpurun_tmp/k_native_head.hip.txt gpu-physics-engine_amd/csrc/k_native.hip
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
echo "last commit"; run
cp /tmp/k_native_new.hip gpu-physics-engine_amd/csrc/k_native.hip
GPE_EXTRA_CXXFLAGS="-DGPE_PAIR_IEEE" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
echo "current, IEEE pair arithmetic"; run
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
