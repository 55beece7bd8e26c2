#!/bin/bash
# Round 4, VERDICT task 5: the 100 M gravity-on scene in its crushed regime.  (1) ms/step at marks, product build;
# (2) pairs walked / resolved per step at the same marks, diagnostic build (gpurun_tmp/variants/pairs.so);
# (3) rocprofv3 kernel trace of the product run, per-kernel time in steps 1200-1300 and 2400-2500.
# usage: bash scripts/gpu_soak_r04.sh [ms] [pairs] [trace]   (default: all three legs)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
L=$ROOT/gpu-physics-engine_amd/libgpe.so
MARKS="60 250 500 750 1000 1250 1500 2000 2450"
LEGS=${*:-ms pairs trace}
cd $ROOT
if [[ " $LEGS " == *" ms "* ]]; then
timeout -k 10 400 python scripts/soak_pairs.py 100000000 100 $MARKS > $OUT/soak_ms_per_step.txt 2>&1 || { echo "soak failed"; tail -5 $OUT/soak_ms_per_step.txt; exit 1; }
tail -12 $OUT/soak_ms_per_step.txt
fi
if [[ " $LEGS " == *" pairs "* ]] && [ -f gpurun_tmp/variants/pairs.so ]; then
  cp $L /tmp/libgpe_default.so; cp gpurun_tmp/variants/pairs.so $L
  timeout -k 10 600 python scripts/soak_pairs.py 100000000 100 $MARKS > $OUT/soak_pairs_per_step.txt 2>&1; rc=$?
  cp /tmp/libgpe_default.so $L
  [ $rc -ne 0 ] && { echo "pairs soak failed"; tail -5 $OUT/soak_pairs_per_step.txt; exit 1; }
  tail -12 $OUT/soak_pairs_per_step.txt
fi
[[ " $LEGS " == *" trace "* ]] || exit 0
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/soak_trace; mkdir -p $OUT/soak_trace
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/soak_trace -- python3 $ROOT/scripts/soak_pairs.py 100000000 100 ${TRACE_MARKS:-1250 2450} > $OUT/soak_trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/soak_trace.log; exit 1; }
f=$(find $OUT/soak_trace -name '*kernel_trace.csv' | head -1)
python3 $ROOT/scripts/trace_windows.py $f ${TRACE_WINDOWS:-10:110 1200:1300 2400:2500} > $OUT/soak_kernels_by_window.txt
rm -rf $OUT/soak_trace
cat $OUT/soak_kernels_by_window.txt
