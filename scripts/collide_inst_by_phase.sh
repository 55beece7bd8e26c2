#!/bin/bash
# VALU / SALU / LDS instruction counts of the collide kernel with phases omitted (diagnostic builds; results wrong)
set -u
N=${1:-16000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2/inst_by_phase
mkdir -p "$OUT"
for skip in ${SKIPS:-0 16 8 1 2 4}; do
  cd "$ROOT"
  GPE_EXTRA_CXXFLAGS="-DGPE_DBG_SKIP=$skip" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || exit 1
  cd /tmp && export TMPDIR=/tmp
  rm -rf "$OUT/tmp"; mkdir -p "$OUT/tmp"
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/tmp" -- python3 "$ROOT/scripts/time_step.py" $N 6 > "$OUT/log_$skip.txt" 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout"; exit 1; fi
  f=$(find "$OUT/tmp" -name '*counter_collection.csv' | head -1)
  echo "skip=$skip" | tee -a "$OUT/summary.txt"
  python3 - "$f" <<'PY' | tee -a "$OUT/summary.txt"
import csv, sys, collections
agg = collections.defaultdict(float); seen=set()
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if "k_collide_dense" not in row["Kernel_Name"]: continue
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); seen.add(row.get("Dispatch_Id"))
n=max(1,len(seen))
print("   " + "  ".join("%s %.1fM" % (k.replace("SQ_",""), v/n/1e6) for k, v in sorted(agg.items())))
PY
  rm -rf "$OUT/tmp"
done
cd "$ROOT"; python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
