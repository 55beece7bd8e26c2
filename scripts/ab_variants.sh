#!/bin/bash
# same box: bench.py (1M) for k_native.hip variants kept under gpurun_tmp/ (diagnostic)
set -u
run() { for r in 1 2; do timeout -k 10 120 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   1M ms/step %.4f  collide %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; done; }
cp gpu-physics-engine_amd/csrc/k_native.hip /tmp/k_native_new.hip
for v in "$@"; do
  cp gpurun_tmp/$v gpu-physics-engine_amd/csrc/k_native.hip
  python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  echo "$v"; run
done
cp /tmp/k_native_new.hip gpu-physics-engine_amd/csrc/k_native.hip
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
