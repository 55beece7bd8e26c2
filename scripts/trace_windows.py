"""Per-kernel time inside step windows of a rocprofv3 kernel trace (csv): python scripts/trace_windows.py <kernel_trace.csv> lo:hi [lo:hi ...]
A step = one k_native_hash dispatch and everything up to the next one (configuration-time launches come before step 0)."""
import collections, csv, sys
path, wins = sys.argv[1], [tuple(int(v) for v in w.split(":")) for w in sys.argv[2:]]
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
step = -1
per = [collections.defaultdict(lambda: [0, 0]) for _ in wins]
span = [[None, None, 0] for _ in wins]
for t0, t1, name in rows:
    if "k_native_hash" in name:
        step += 1
    for i, (lo, hi) in enumerate(wins):
        if lo <= step < hi:
            per[i][name][0] += t1 - t0; per[i][name][1] += 1
            if span[i][0] is None: span[i][0] = t0
            span[i][1] = t1
for i, (lo, hi) in enumerate(wins):
    n = hi - lo
    wall = (span[i][1] - span[i][0]) / 1e6 / n if span[i][0] else 0.0
    busy = sum(v[0] for v in per[i].values()) / 1e6 / n
    print("steps %d-%d: %.3f ms/step wall (first kernel start to last kernel end), %.3f ms/step inside kernels" % (lo, hi, wall, busy))
    for name, (ns, cnt) in sorted(per[i].items(), key=lambda kv: -kv[1][0]):
        print("   %-64s %9.3f ms/step   %6.2f launches/step   %9.1f us/launch" % (name[:64], ns / 1e6 / n, cnt / n, ns / 1e3 / max(1, cnt)))
