"""Per-launch durations of one training step from a rocprofv3 kernel trace: python tools/train_layers.py <kernel_trace.csv> [kernel substring]
Groups the launches of the LAST complete step by (kernel, grid size) and prints time per group - which layers a kernel family spends its time on."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "hcol_reduce"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step = the launches between two loss kernels
idx = [i for i, r in enumerate(rows) if "loss_kernel" in r["Kernel_Name"]]
lo, hi = idx[-3], idx[-2]
agg = collections.OrderedDict()
for r in rows[lo:hi]:
    if pat not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:60], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
print("step total for '%s': %.1f us in %d launches" % (pat, tot, sum(a[0] for a in agg.values())))
for (k, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-62s blocks %6d  x%3d  avg %7.1f us  sum %8.1f us" % (k, g, n, t / n, t))

seq = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[lo:hi] if pat in r["Kernel_Name"]]
print("in launch order (us):", " ".join("%.0f" % d for d in seq))
