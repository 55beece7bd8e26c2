#!/bin/bash
# VALU / SALU / LDS instruction counts and cycles of the direct-slot collide kernel with phases omitted, one library per
# omission mask (built here: for k in 0 1 3 7 39 103 8 16; do bash scripts/build_variant.sh skip$k "-DGPE_DBG_SKIP=$k"; done;
# results of those builds are WRONG, only the counters are of interest).  Masks: 1 colour passes, 2 P4 lists, 4 filing,
# 8 lane groups only, 16 one-lane cells only, 32 write-back + prev prefetch, 64 the gather loop.
set -u
N=${1:-16000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04/inst_by_phase
mkdir -p "$OUT"
L=$ROOT/gpu-physics-engine_amd/libgpe.so
cp $L /tmp/libgpe_default.so
cd /tmp && export TMPDIR=/tmp
for skip in ${SKIPS:-0 1 3 7 39 103 8 16}; do
  cp $ROOT/gpurun_tmp/variants/skip$skip.so $L || exit 1
  rm -rf "$OUT/tmp"; mkdir -p "$OUT/tmp"
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/tmp" -- python3 "$ROOT/scripts/time_step.py" $N 6 > "$OUT/log_$skip.txt" 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout"; cp /tmp/libgpe_default.so $L; exit 1; fi
  f=$(find "$OUT/tmp" -name '*counter_collection.csv' | head -1)
  echo "skip=$skip" | tee -a "$OUT/summary.txt"
  python3 - "$f" <<'PY' | tee -a "$OUT/summary.txt"
import csv, sys, collections
agg = collections.defaultdict(float); seen=set()
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        if "k_collide_direct" not in row["Kernel_Name"]: continue
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); seen.add(row.get("Dispatch_Id"))
n=max(1,len(seen))
print("   dispatches %d  " % n + "  ".join("%s %.2fM" % (k.replace("SQ_",""), v/n/1e6) for k, v in sorted(agg.items())))
PY
  rm -rf "$OUT/tmp"
done
cp /tmp/libgpe_default.so $L
