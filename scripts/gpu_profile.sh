#!/bin/bash
# rocprofv3 on the GPU box: kernel-trace stats of bench.py, then (separate runs) PMC passes for HBM bytes.
# usage: bash scripts/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace =="
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/trace_bench.log" 2>&1
rc=$?; echo "trace rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
grep -h '^{' "$OUT/trace_bench.log" | tail -1 > "$OUT/bench_under_rocprof.json"
find "$OUT/trace" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
head -30 "$OUT/kernel_stats.csv"
if [ "${PMC:-1}" = "1" ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $ctr =="
    timeout -k 10 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/pmc_$ctr" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extra --steps 5 --warmup 2 "$@" > "$OUT/pmc_${ctr}_bench.log" 2>&1
    rc=$?; echo "pmc $ctr rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
    f=$(find "$OUT/pmc_$ctr" -name '*counter_collection.csv' | head -1)
    python3 - "$f" "$ctr" > "$OUT/pmc_${ctr}_summary.txt" <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
with open(f) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") != ctr: continue
        name = row["Kernel_Name"].split("(")[0]
        agg[name][0] += float(row["Counter_Value"]); agg[name][1] += 1
print("kernel,%s_sum,dispatches,%s_per_dispatch" % (ctr, ctr))
for k, (v, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%s,%.1f,%d,%.1f" % (k, v, c, v / max(1, c)))
PY
    head -20 "$OUT/pmc_${ctr}_summary.txt"
    rm -rf "$OUT/pmc_$ctr"
  done
fi
# keep the merge small
find "$OUT/trace" -name '*kernel_trace.csv' -size +20M -delete
ls -la "$OUT"
