"""Does the exchange's kernel run beside the tiles?  From a rocprofv3 kernel trace of scripts/shard_overhead_one_rank.py:
for every RCCL kernel of the steady state, where it starts and ends relative to the collide launch it should overlap.
python scripts/trace_overlap.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Stream_Id", "?")))
rows.sort()
rows = rows[len(rows) // 2:]
coll = [r for r in rows if "k_collide_direct" in r[2]]
nccl = [r for r in rows if "nccl" in r[2]]
unp = [r for r in rows if "k_shard_unpack" in r[2]]
inside = 0; lead = []; gap_unpack = []
for s, e, _, _ in nccl:
    c = [x for x in coll if x[0] <= e and x[1] >= s]
    if c: inside += 1
    prev = [x for x in coll if x[0] <= s]
    if prev: lead.append((s - prev[-1][0]) / 1e3)
    nxt = [x for x in unp if x[0] >= e]
    if nxt: gap_unpack.append((nxt[0][0] - e) / 1e3)
print("%d RCCL kernels, %d of them overlap a collide launch in time; start %.1f us after the collide launch's start (mean); "
      "the unpack starts %.1f us after the RCCL kernel's end (mean)" % (len(nccl), inside, sum(lead) / max(1, len(lead)), sum(gap_unpack) / max(1, len(gap_unpack))))
span = rows[-1][1] - rows[0][0]
steps = len(unp)
print("steady state: %.1f us per step over %d steps" % (span / 1e3 / max(1, steps), steps))
import collections
agg = collections.OrderedDict()
for s, e, k, st in rows:
    a = agg.setdefault((k, st), [0, 0]); a[0] += 1; a[1] += e - s
for (k, st), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print("   %-50s stream %-4s %6d launches %8.2f us each" % (k, st, c, d / c / 1e3))
