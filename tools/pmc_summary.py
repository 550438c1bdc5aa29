#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (CSV) per kernel symbol and grid size.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [--json out.json]
Units: the counters are in KiB.  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md §HBM), so `hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024`."""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "ynk::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("void ynk::", "").replace("ynk::", "")
        name = name[:name.index("(")] if "(" in name else name
        agg[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    print("| kernel | grid (threads) | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM MB/launch = (2*F+W)*1024 |\n|---|---|---|---|---|")
    tot = {}
    for k in sorted(set(f) | set(w)):
        (fk, nf), (wk, nw) = f.get(k, (0.0, 0)), w.get(k, (0.0, 0))
        mb = (2 * fk + wk) * 1024 / 1e6
        out.setdefault(k[0], {})[str(k[1])] = {"fetch_kib": fk, "write_kib": wk, "hbm_mb": mb, "samples": max(nf, nw)}
        t = tot.setdefault(k[0], [0.0, 0])
        t[0] += mb * max(nf, nw)
        t[1] += max(nf, nw)
        print("| `%s` | %d | %.0f | %.0f | %.1f |" % (k[0], k[1], fk, wk, mb))
    for name, (m, n) in tot.items():
        out[name]["avg_hbm_mb_per_launch"] = m / max(n, 1)
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
