#!/bin/bash
# same box: bench.py for the working tree vs the csrc/ of the last commit (gpurun_tmp/csrc_head/, made by the caller:
#   rm -rf gpurun_tmp/csrc_head && mkdir -p gpurun_tmp/csrc_head && for f in $(git ls-files gpu-physics-engine_amd/csrc); do git show HEAD:$f > gpurun_tmp/csrc_head/$(basename $f); done)
set -u
run() { for r in 1 2 3; do timeout -k 10 120 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   1M ms/step %.4f  collide %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; done
timeout -k 10 200 python bench.py --particles 100000000 --gravity on --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   100M ms/step %.4f  collide %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
echo "working tree"; python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1; run
mkdir -p /tmp/csrc_new && cp gpu-physics-engine_amd/csrc/*.hip gpu-physics-engine_amd/csrc/*.h /tmp/csrc_new/
cp gpurun_tmp/csrc_head/* gpu-physics-engine_amd/csrc/
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
echo "last commit"; run
cp /tmp/csrc_new/* gpu-physics-engine_amd/csrc/
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
