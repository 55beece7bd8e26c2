#!/bin/bash
# The bench headline (20 windows of 100 steps following each other at 1 M) for several prebuilt libraries, twice each,
# interleaved, same box: value, the first / median / slowest window.   usage: bash scripts/ab_long_run.sh <tag> <name> ...
set -u
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
L=$ROOT/gpu-physics-engine_amd/libgpe.so; cp $L /tmp/libgpe_default.so
cd $ROOT; : > $OUT/ab_long_run_$tag.txt
for r in 1 2; do
  for v in "$@"; do
    cp gpurun_tmp/variants/$v.so $L || { echo "no variant $v"; continue; }
    timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=j['timing']
print('[$v] steps/s %.0f  ms/step %.5f  fresh %.0f  windows (ms per 100 steps): first %.2f median %.2f max %.2f  collide %.4f ms' % (j['value'], j['ms_per_step'], j['value_fresh_cloud'], t['first_window_ms'], t['median_window_ms'], t['max_window_ms'], j['roofline']['avg_launch_ms']))" | tee -a $OUT/ab_long_run_$tag.txt
  done
done
cp /tmp/libgpe_default.so $L
