#!/usr/bin/env python3
"""Per-kernel summary of an SQ counter pass:
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
            SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d out -- python3 bench.py ...
  python tools/sq_summary.py out/*/*_counter_collection.csv
Fractions are of SQ_WAVE_CYCLES (wave-resident quad-cycles): parked = waiting on s_waitcnt / barriers, stall = issue stalls,
active = issuing; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CYCLES-ish) is left raw."""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "ynk::" not in n:
        continue
    n = n.replace("void ynk::", "").replace("ynk::", "")
    n = n[:n.index("(")] if "(" in n else n
    agg[(n, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-44s %9s %6s | %7s %7s %7s | %9s %9s %9s" % ("kernel", "grid", "calls", "parked", "stall", "active", "mfma_busy", "lds_act", "lds_conf"))
out = []
for (n, g), c in agg.items():
    med = {k: sorted(v)[len(v) // 2] for k, v in c.items()}
    wc = med.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    out.append((wc * len(c.get("SQ_WAVE_CYCLES", [1])), n, g, len(c.get("SQ_WAVE_CYCLES", [])), med, wc))
for tot, n, g, calls, med, wc in sorted(out, reverse=True)[:40]:
    print("%-44s %9s %6d | %6.1f%% %6.1f%% %6.1f%% | %9.3g %9.3g %9.3g" % (
        n[:44], g, calls, 100 * med.get("SQ_WAIT_ANY", 0) / wc, 100 * med.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * med.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        med.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), med.get("SQ_LDS_IDX_ACTIVE", 0), med.get("SQ_LDS_BANK_CONFLICT", 0)))
