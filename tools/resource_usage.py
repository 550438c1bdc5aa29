#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr of a compile):
   python tools/resource_usage.py new.txt [old.txt]  -> one line per kernel, and what changed against old.txt"""
import re
import subprocess
import sys


def parse(path):
    out, cur = {}, None
    for ln in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize|Occupancy|LDS Size)[^:]*: (\d+)", ln)
        if m and cur:
            out[cur][m.group(1)] = int(m.group(2))
    return out


def dem(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("ynk::", "")[:70]
    except Exception:
        return n


new = parse(sys.argv[1])
old = parse(sys.argv[2]) if len(sys.argv) > 2 else {}
for k, v in sorted(new.items()):
    o = old.get(k)
    flag = ""
    if o and (o.get("VGPRs") != v.get("VGPRs") or o.get("Occupancy") != v.get("Occupancy") or o.get("ScratchSize") != v.get("ScratchSize")):
        flag = "   <-- was VGPR %s occ %s scratch %s" % (o.get("VGPRs"), o.get("Occupancy"), o.get("ScratchSize"))
    if len(sys.argv) > 2 and not flag:
        continue
    print("%-72s VGPR %3s AGPR %3s SGPR %3s scratch %3s occ %s%s" % (dem(k), v.get("VGPRs"), v.get("AGPRs"), v.get("TotalSGPRs"), v.get("ScratchSize"), v.get("Occupancy"), flag))
