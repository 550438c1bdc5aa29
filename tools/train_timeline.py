"""One training step from a rocprofv3 kernel trace, in launch order per queue: python tools/train_timeline.py <kernel_trace.csv>
For every launch of the last complete step: queue, kernel, workgroups, duration, and the idle gap on its queue since the previous
launch ended.  Ends with per-queue totals (busy, gaps, span) - where the step's wall time goes."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "loss_kernel" in r["Kernel_Name"]]
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
# a step: from the first launch after the previous sgd to this step's sgd
hi = sgd[-2]
lo = sgd[-3] + 1
step = rows[lo:hi + 1]
t0 = int(step[0]["Start_Timestamp"])
last_end = {}
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
short = lambda n: re.sub(r"\(.*", "", n.replace("ynk::", "").replace("void ", ""))[:44]
qnames = {}
for r in step:
    q = r.get("Queue_Id", "0")
    qnames.setdefault(q, "Q%d" % len(qnames))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    blocks = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    tot[q][0] += (e - s) / 1e3; tot[q][1] += max(gap, 0.0); tot[q][2] += 1
    print("%s %9.1f  %-44s wg %6d  dur %7.1f  gap %6.1f" % (qnames[q], (s - t0) / 1e3, short(r["Kernel_Name"]), blocks, (e - s) / 1e3, gap))
span = (max(int(r["End_Timestamp"]) for r in step) - t0) / 1e3
print("step span %.1f us" % span)
for q, (busy, gaps, n) in tot.items():
    print("%s: %d launches, busy %.1f us, gaps %.1f us" % (qnames[q], n, busy, gaps))
