#!/usr/bin/env python3
"""What happens when the 4 streams of the default bench share the GPU: joins a rocprofv3 kernel trace of the DEFAULT run (four
handles / four streams) with one of the ONE-stream run and prints, per kernel symbol, the average duration alone and under
co-residency (inflation), its per-workgroup resources and how many workgroups of it a CU holds; plus the time-weighted number of
kernels in flight.  No counters involved (a --pmc pass serialises the dispatches, so it cannot see contention; the timestamps of the
plain trace can).

The per-workgroup resources do NOT come from the trace (round 4's table did, and was wrong where it mattered: the trace's LDS_Block_Size
is the STATIC allocation only - 0 for every kernel that sizes its LDS at launch - and its VGPR_Count leaves the accumulation registers
out).  They come from
  * the compiler's own report (hipcc -Rpass-analysis=kernel-resource-usage, parsed by tools/resource_usage.py's reader): VGPRs + AGPRs,
    static LDS, scratch;
  * the library's launch log (YN_LOG_LDS=1: "yn_lds <kernel> <dynamic bytes> <threads>", one line per distinct launch shape).

usage: python tools/concurrency.py <default_kernel_trace.csv> <one_stream_kernel_trace.csv> <resource_usage.txt> <dynamic_lds.txt>"""
import collections
import csv
import re
import subprocess
import sys


def norm(n):
    n = n.replace("void ynk::", "").replace("ynk::", "").strip().strip("()")
    n = n[:n.index("(")] if "(" in n else n
    return re.sub(r"\s+", "", n)


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "ynk::" not in n:
            continue
        wg = int(r.get("Workgroup_Size_X", 256) or 256) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
        grid = int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), norm(n), wg, grid, r.get("Queue_Id", "?")))
    rows.sort()
    return rows


def steady(rows, skip=0.5):
    """the last (1-skip) of the trace: past autotuning and warm-up"""
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + (t1 - t0) * skip
    return [r for r in rows if r[0] >= cut]


def compiler_table(path):
    """kernel -> {VGPRs (incl. AGPRs), LDS (static), scratch} from the -Rpass-analysis=kernel-resource-usage remarks"""
    out, cur = {}, None
    names = []
    for ln in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            names.append(cur)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize|LDS Size)[^:]*: (\d+)", ln)
        if m and cur:
            out[cur][m.group(1)] = int(m.group(2))
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
    return {norm(d): out[n] for n, d in zip(names, dem)}


def launch_table(path):
    """kernel -> largest dynamic LDS it was launched with"""
    out = {}
    for ln in open(path, errors="replace"):
        f = ln.split()
        if len(f) >= 4 and f[0] == "yn_lds":
            k = norm(" ".join(f[1:-2]))
            out[k] = max(out.get(k, 0), int(f[-2]))
    return out


def main():
    multi, single = steady(load(sys.argv[1])), steady(load(sys.argv[2]))
    comp = compiler_table(sys.argv[3]) if len(sys.argv) > 3 else {}
    dyn = launch_table(sys.argv[4]) if len(sys.argv) > 4 else {}
    solo = collections.defaultdict(list)
    for s, e, n, *_ in single:
        solo[n].append((e - s) / 1e3)
    agg = collections.defaultdict(lambda: {"d": [], "res": None})
    for s, e, n, wg, grid, q in multi:
        agg[n]["d"].append((e - s) / 1e3)
        agg[n]["res"] = (wg, grid)
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
    print("## kernels in flight (default run, steady-state half of the trace; HSA queues used: %s)\n" % sorted({r[5] for r in multi}))
    print("time-weighted mean %.2f; " % (busy / span) + ", ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
    # per HSA queue (= bench.py stream): how much of the span it has a kernel executing.  A queue that is busy well under 100 % is waiting for
    # its launches, not for the GPU: rocprofv3's tracing multiplies the cost of a launch, so THIS run is bound by the launching thread and
    # holds fewer kernels in flight than the unprofiled run (tools/probe/enqueue_time.py: 120 us to enqueue a batch that takes 1 050 us;
    # tools/probe/queue_probe.hip: four streams do run four kernels at once on this GPU)
    print("\nper queue: " + ", ".join("%s busy %.0f %%" % (q, 100.0 * sum(e - s for s, e, *r in multi if r[-1] == q) / span) for q in sorted({r[5] for r in multi})))
    print("\n## per kernel symbol (registers / LDS: compiler report + launch log, see the docstring)\n")
    print("| kernel | calls | avg us alone (1 stream) | avg us in the 4-stream run | inflation | LDS B/workgroup (static + dynamic) | VGPRs + AGPRs | threads | workgroups | workgroups/CU (registers / LDS / wave slots) | share of the chip's workgroup slots it fills alone |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    tot_m = sum(sum(v["d"]) for v in agg.values())
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["d"])):
        wg, grid = v["res"]
        c = comp.get(n, {})
        regs = c.get("VGPRs", 0) + c.get("AGPRs", 0)
        lds = c.get("LDS Size", 0) + dyn.get(n, 0)
        blocks = grid // max(wg, 1)
        waves = max(wg // 64, 1)
        alloc = ((max(regs, 1) + 7) // 8) * 8
        by_vgpr = (min(8, 512 // alloc) * 4) // waves if regs else 0
        by_lds = (160 * 1024) // lds if lds else 32
        by_waves = 32 // waves
        per_cu = max(1, min(by_vgpr or 99, by_lds, by_waves))
        a = sum(v["d"]) / len(v["d"])
        s1 = solo.get(n)
        s_avg = sum(s1) / len(s1) if s1 else float("nan")
        print("| `%s` | %d | %.1f | %.1f | %.2f | %d | %s | %d | %d | %d (%s / %d / %d) | %.2f |" % (
            n[:48], len(v["d"]), s_avg, a, a / s_avg if s1 else float("nan"), lds, regs if regs else "?", wg, blocks, per_cu, by_vgpr if regs else "?", by_lds, by_waves,
            min(1.0, blocks / (256.0 * per_cu))))
    print("\nsum of kernel durations / wall span = %.2f" % (tot_m * 1e3 / span))


if __name__ == "__main__":
    main()
