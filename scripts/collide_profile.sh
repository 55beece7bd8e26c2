#!/bin/bash
# usage: bash scripts/collide_profile.sh <tag> [N]   (GPU box; writes gpurun_out/<tag>/)
# SQ counter passes on the collide kernel (program after `--`, no wrappers), then phase costs by omission.
set -u
TAG=${1:-collide}; N=${2:-16000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters
  local name=$1; shift
  rm -rf "$OUT/pmc_tmp"; mkdir -p "$OUT/pmc_tmp"
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_tmp" -- python3 "$ROOT/scripts/time_step.py" $N 6 > "$OUT/pmc_$name.log" 2>&1
  local rc=$?; echo "pmc $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  f=$(find "$OUT/pmc_tmp" -name '*counter_collection.csv' | head -1)
  python3 - "$f" > "$OUT/pmc_$name.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen=set(); cnt=collections.Counter()
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row["Kernel_Name"].split("(")[0]
        if "collide" not in name and "os_pass" not in name and "native_hash" not in name: continue
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
        key=(name,row.get("Dispatch_Id"))
        if key not in seen: seen.add(key); cnt[name]+=1
for k, d in agg.items():
    print(k, "dispatches", cnt[k])
    for c, v in sorted(d.items()): print("   %-28s per-dispatch %18.1f" % (c, v / max(1, cnt[k])))
PY
  cat "$OUT/pmc_$name.txt"; rm -rf "$OUT/pmc_tmp"
}
pass inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
cd "$ROOT"
if [ "${SKIPS:-}" != "none" ]; then
  for skip in ${SKIPS:-0 16 8 24 1 2 4}; do
    GPE_EXTRA_CXXFLAGS="-DGPE_DBG_SKIP=$skip" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || exit 1
    echo "skip=$skip" | tee -a "$OUT/skips.txt"; timeout -k 10 120 python scripts/time_step.py $N 40 2>&1 | tail -1 | cut -c1-160 | tee -a "$OUT/skips.txt"
  done
  python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1
fi
