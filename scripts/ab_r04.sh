#!/bin/bash
# Round 4 A/B on one box: for each prebuilt library (scripts/build_variant.sh, or a copy of libgpe.so under
# gpurun_tmp/variants/) the native step's scopes at 1 M (twice, interleaved), 16 M and 100 M, then the collide kernel's
# instruction counters at 16 M (SQ_INSTS_VALU per launch x 64 / N = VALU lane-slots per particle).
# usage: bash scripts/ab_r04.sh [-q] [-t] <tag> <name> ...     -q: no 100 M leg;  -t: parity tests with every library first
set -u
quick=0; tests=0
while [ "${1:-}" = "-q" ] || [ "${1:-}" = "-t" ]; do [ "$1" = "-q" ] && quick=1; [ "$1" = "-t" ] && tests=1; shift; done
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
L=$ROOT/gpu-physics-engine_amd/libgpe.so
LOG=$OUT/ab_$tag.txt; : > $LOG
cp $L /tmp/libgpe_default.so
restore() { cp /tmp/libgpe_default.so $L; }
cd $ROOT
if [ $tests -eq 1 ]; then
  for v in "$@"; do
    cp gpurun_tmp/variants/$v.so $L || { echo "no variant $v"; restore; exit 1; }
    echo "[$v] parity tests" | tee -a $LOG
    timeout -k 10 900 python -m pytest tests/test_gpu_native.py tests/test_gpu_parity_step.py -m gpu -x -q --timeout 600 2>&1 | tail -3 | tee -a $LOG
    rc=${PIPESTATUS[0]}; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout"; restore; exit 1; fi
  done
fi
for r in 1 2; do
  for v in "$@"; do
    cp gpurun_tmp/variants/$v.so $L || { echo "no variant $v"; restore; exit 1; }
    echo -n "[$v] " | tee -a $LOG; timeout -k 10 120 python scripts/time_step.py 1000000 400 2>&1 | grep "^n=" | cut -c1-260 | tee -a $LOG
  done
done
for v in "$@"; do
  cp gpurun_tmp/variants/$v.so $L
  echo -n "[$v] " | tee -a $LOG; timeout -k 10 120 python scripts/time_step.py 16000000 60 2>&1 | grep "^n=" | cut -c1-260 | tee -a $LOG
  [ $quick -eq 0 ] && { echo -n "[$v] " | tee -a $LOG; timeout -k 10 240 python scripts/time_step.py 100000000 30 on 2>&1 | grep "^n=" | cut -c1-260 | tee -a $LOG; }
done
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $ROOT/gpurun_tmp/variants/$v.so $L
  rm -rf $OUT/pmc_tmp; mkdir -p $OUT/pmc_tmp
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_tmp -- python3 $ROOT/scripts/time_step.py 16000000 6 > $OUT/pmc_tmp.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout"; restore; exit 1; fi
  f=$(find $OUT/pmc_tmp -name '*counter_collection.csv' | head -1)
  echo -n "[$v] 16 M " | tee -a $LOG
  python3 - "$f" <<'PY' | tee -a $LOG
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row["Kernel_Name"].split("(")[0].replace("void gpe::", "")
        if "k_collide_direct" not in name and "k_native_hash" not in name: continue
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"]); seen[name].add(row.get("Dispatch_Id"))
for name, d in sorted(agg.items()):
    n = max(1, len(seen[name]))
    print("%s: dispatches %d  " % (name[:40], n) + "  ".join("%s %.2fM" % (k.replace("SQ_", ""), v / n / 1e6) for k, v in sorted(d.items())) +
          "  => %.0f VALU lane-slots per particle" % (d["SQ_INSTS_VALU"] / n * 64 / 16e6))
PY
  rm -rf $OUT/pmc_tmp
done
restore
