#!/bin/bash
# usage: bash scripts/gpu_pmc.sh "<counters...>" <kernel-substring> [bench args...]
# One rocprofv3 --pmc pass (counters must fit one pass), per-kernel sums printed.
set -u
CTRS=$1; KSUB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tmp
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 4 --warmup 2 "$@" > "$OUT/bench.log" 2>&1
echo "rc=$?"
f=$(find "$OUT" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$KSUB" <<'PY'
import csv, sys, collections
f, ksub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen=set()
with open(f) as fh:
    for row in csv.DictReader(fh):
        name = row["Kernel_Name"].split("(")[0]
        if ksub not in name: continue
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
        key=(name,row.get("Dispatch_Id"))
        if key not in seen: seen.add(key); cnt[name]+=1
for k, d in agg.items():
    print(k, "dispatches", cnt[k])
    for c, v in sorted(d.items()): print("   %-28s %18.1f  per-dispatch %16.1f" % (c, v, v / max(1, cnt[k])))
PY
rm -rf "$OUT"
