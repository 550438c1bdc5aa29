#!/usr/bin/env python3
"""What happens when the 4 streams of the default bench share the GPU: joins a rocprofv3 kernel trace of the DEFAULT run (four
handles / four streams) with one of the ONE-stream run and prints, per kernel symbol, the average duration alone and under
co-residency (inflation), its per-block resources (LDS, VGPRs, waves) and how much of a CU's residency the launch takes on its own;
plus the time-weighted number of kernels in flight.  No counters involved (a --pmc pass serialises the dispatches, so it cannot see
contention; the timestamps of the plain trace can).

usage: python tools/concurrency.py <default_kernel_trace.csv> <one_stream_kernel_trace.csv> [steps_to_skip_fraction]"""
import collections
import csv
import sys


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "ynk::" not in n:
            continue
        n = n.replace("void ynk::", "").replace("ynk::", "")
        n = n[:n.index("(")] if "(" in n else n
        wg = int(r.get("Workgroup_Size_X", 256) or 256) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        grid = int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("LDS_Block_Size", 0) or 0),
                     int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0), wg, grid, r.get("Queue_Id", "?")))
    rows.sort()
    return rows


def steady(rows, skip=0.5):
    """the last (1-skip) of the trace: past autotuning and warm-up"""
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + (t1 - t0) * skip
    return [r for r in rows if r[0] >= cut]


def main():
    multi, single = steady(load(sys.argv[1])), steady(load(sys.argv[2]))
    solo = collections.defaultdict(list)
    for s, e, n, *_ in single:
        solo[n].append((e - s) / 1e3)
    agg = collections.defaultdict(lambda: {"d": [], "res": None})
    for s, e, n, lds, vg, wg, grid, q in multi:
        agg[n]["d"].append((e - s) / 1e3)
        agg[n]["res"] = (lds, vg, wg, grid)
    # kernels in flight over time (sweep)
    ev = []
    for s, e, *_ in multi:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    hist, cur, last = collections.Counter(), 0, ev[0][0]
    for t, d in ev:
        hist[cur] += t - last
        cur += d; last = t
    span = sum(hist.values())
    busy = sum((e - s) for s, e, *_ in multi)
    print("## kernels in flight (default run, steady-state half of the trace; HSA queues used: %s)\n" % sorted({r[7] for r in multi}))
    print("time-weighted mean %.2f; " % (busy / span) + ", ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
    print("\n## per kernel symbol\n")
    print("| kernel | calls | avg us alone (1 stream) | avg us in the 4-stream run | inflation | LDS B/block | VGPRs | blocks | blocks/CU its resources allow | share of the chip's block slots it fills alone |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    tot_m = sum(sum(v["d"]) for v in agg.values())
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["d"])):
        lds, vg, wg, grid = v["res"]
        blocks = grid // max(wg, 1)
        waves = max(wg // 64, 1)
        alloc = ((max(vg, 1) + 7) // 8) * 8
        by_vgpr = (min(8, 512 // alloc) * 4) // waves if alloc else 8
        by_lds = (160 * 1024) // lds if lds else 32
        by_waves = 32 // waves
        per_cu = max(1, min(by_vgpr, by_lds, by_waves, 16))
        a = sum(v["d"]) / len(v["d"])
        s1 = solo.get(n)
        s_avg = sum(s1) / len(s1) if s1 else float("nan")
        print("| `%s` | %d | %.1f | %.1f | %.2f | %d | %d | %d | %d | %.2f |" % (n[:48], len(v["d"]), s_avg, a, a / s_avg if s1 else float("nan"), lds, vg, blocks, per_cu,
                                                                        min(1.0, blocks / (256.0 * per_cu))))
    print("\nsum of kernel durations / wall span = %.2f" % (tot_m * 1e3 / span))


if __name__ == "__main__":
    main()
