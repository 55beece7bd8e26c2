"""Per-kernel duration and idle gap before each kernel, from a rocprofv3 kernel_trace.csv (steady-state tail)."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
tail = rows[len(rows) // 2:]
agg = collections.OrderedDict()
prev_end = None
for s, e, k in tail:
    a = agg.setdefault(k, [0, 0, 0])
    a[0] += 1; a[1] += e - s
    if prev_end is not None: a[2] += max(0, s - prev_end)
    prev_end = max(prev_end or e, e)
span = tail[-1][1] - tail[0][0]
busy = sum(a[1] for a in agg.values())
print("span %.3f ms  busy %.3f ms (%.1f%%)" % (span / 1e6, busy / 1e6, 100.0 * busy / span))
print("%-60s %8s %10s %10s" % ("kernel", "calls", "avg us", "gap us"))
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-60s %8d %10.2f %10.2f" % (k, c, d / c / 1e3, g / c / 1e3))
