#!/bin/bash
# Round 4, VERDICT task 5: the crushed regime of the 100 M gravity-on scene under rocprofv3 -- kernel trace, then three
# separate PMC passes (FETCH_SIZE; WRITE_SIZE; SQ instruction counters), each over a whole run to step 2500, summarised for
# the step windows 10-110, 1200-1300 and 2400-2500 -> gpurun_out/r04/soak_final_*.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
W="10:110 1200:1300 2400:2500"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sk; mkdir -p /tmp/sk
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/sk -- python3 $ROOT/scripts/soak_pairs.py 100000000 100 1250 2450 > $OUT/soak_final_trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/soak_final_trace.log; exit 1; }
python3 $ROOT/scripts/trace_windows.py $(find /tmp/sk -name '*kernel_trace.csv' | head -1) $W > $OUT/soak_final_kernels_by_window.txt
cat $OUT/soak_final_kernels_by_window.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/sk; mkdir -p /tmp/sk
  timeout -k 10 500 rocprofv3 --pmc $pass --output-format csv -d /tmp/sk -- python3 $ROOT/scripts/soak_pairs.py 100000000 100 1250 2450 > $OUT/soak_final_pmc_$name.log 2>&1
  rc=$?; echo "pmc $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 $ROOT/scripts/pmc_windows.py $(find /tmp/sk -name '*counter_collection.csv' | head -1) $W > $OUT/soak_final_pmc_$name.txt
  head -12 $OUT/soak_final_pmc_$name.txt | cut -c1-200
done
rm -rf /tmp/sk
