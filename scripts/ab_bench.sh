#!/bin/bash
# same box: bench.py (1M, and 16M) with the current k_native.hip vs the one of a reference commit (gpurun_tmp/)
set -u
run() { for r in 1 2 3; do timeout -k 10 120 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   1M ms/step %.4f  collide %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; done
timeout -k 10 120 python bench.py --particles 16000000 --steps 40 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   16M ms/step %.4f  collide %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
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
