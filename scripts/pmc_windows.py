"""Per-kernel counter sums inside step windows of a rocprofv3 --pmc run (counter_collection.csv):
python scripts/pmc_windows.py <counter_collection.csv> lo:hi [lo:hi ...]
A step = one k_native_hash dispatch and everything up to the next one (as scripts/trace_windows.py)."""
import collections, csv, sys
path, wins = sys.argv[1], [tuple(int(v) for v in w.split(":")) for w in sys.argv[2:]]
rows = collections.OrderedDict()
with open(path) as fh:
    for r in csv.DictReader(fh):
        d = int(r["Dispatch_Id"])
        e = rows.setdefault(d, [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpe::", "")[:44], {}])
        e[1][r["Counter_Name"]] = e[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
step = -1
per = [collections.defaultdict(lambda: [0, collections.defaultdict(float)]) for _ in wins]
for d in sorted(rows):
    name, ctr = rows[d]
    if "k_native_hash" in name: step += 1
    for i, (lo, hi) in enumerate(wins):
        if lo <= step < hi:
            a = per[i][name]; a[0] += 1
            for k, v in ctr.items(): a[1][k] += v
for i, (lo, hi) in enumerate(wins):
    print("steps %d-%d (per launch):" % (lo, hi))
    for name, (cnt, ctr) in sorted(per[i].items(), key=lambda kv: -sum(kv[1][1].values())):
        if cnt == 0: continue
        print("   %-46s %6.2f launches/step  " % (name, cnt / (hi - lo)) + "  ".join("%s %.4g" % (k.replace("SQ_", ""), v / cnt) for k, v in sorted(ctr.items())))
