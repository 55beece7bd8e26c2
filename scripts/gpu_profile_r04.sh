#!/bin/bash
# Round-4 profile set (the same passes as gpu_profile_r03.sh) (GPU box): rocprofv3 kernel stats of the default bench, then PMC passes (separate runs, program
# after `--`): FETCH_SIZE, WRITE_SIZE, SQ instruction / wait counters, at 1M (default bench) and 100M.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
summ() {  # csv, out
python3 - "$1" > "$2" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen=set(); cnt=collections.Counter()
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row["Kernel_Name"].split("(")[0]
        if not any(k in name for k in ("collide", "os_pass", "native_hash")): continue
        agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
        key=(name,row.get("Dispatch_Id"))
        if key not in seen: seen.add(key); cnt[name]+=1
for k, d in sorted(agg.items()):
    print(k, "dispatches", cnt[k])
    for c, v in sorted(d.items()): print("   %-28s per-dispatch %20.1f" % (c, v / max(1, cnt[k])))
PY
}
for cfg in "1M --steps 100 --warmup 10 --no-extra --single-window" "100M --particles 100000000 --gravity on --steps 20 --warmup 5 --no-extra --single-window"; do
  tag=${cfg%% *}; args=${cfg#* }
  echo "== kernel trace $tag =="
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$tag" -- python3 "$ROOT/bench.py" --no-cpu-baseline $args > "$OUT/trace_${tag}_bench.log" 2>&1
  rc=$?; echo "trace rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  grep -h '^{' "$OUT/trace_${tag}_bench.log" | tail -1 > "$OUT/bench_under_rocprof_$tag.json"
  find "$OUT/trace_$tag" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$tag.csv"
  head -12 "$OUT/kernel_stats_$tag.csv"
  rm -rf "$OUT/trace_$tag"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    rm -rf "$OUT/pmc_tmp"; mkdir -p "$OUT/pmc_tmp"
    timeout -k 10 400 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_tmp" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 5 --warmup 2 $args > "$OUT/pmc_${tag}_${name}.log" 2>&1
    rc=$?; echo "pmc $tag $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
    f=$(find "$OUT/pmc_tmp" -name '*counter_collection.csv' | head -1)
    summ "$f" "$OUT/pmc_${tag}_${name}.txt"
    rm -rf "$OUT/pmc_tmp"
  done
done
ls -la "$OUT"
