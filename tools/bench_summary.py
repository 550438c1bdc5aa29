#!/usr/bin/env python3
"""Human-readable digest of one bench.py JSON line:  python tools/bench_summary.py gpurun_out/x_bench.json"""
import json
import sys

import os
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("contract line: %d bytes" % len(json.dumps(d)))
if d.get("detail_file"):                                    # round 5: the blocks beyond the contract keys live in bench_detail.json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cand in (d["detail_file"], os.path.join(root, d["detail_file"])):
        if os.path.exists(cand):
            d = dict(json.load(open(cand)), **{k: v for k, v in d.items() if k in ("summary",)})
            d["roofline"] = d.get("roofline_detail") or d.get("roofline")
            break
if d.get("summary"):
    print("summary:", json.dumps(d["summary"]))
print("%s: %.1f %s  (%.4f ms/step)  dtype=%s" % (d["metric"], d["value"], d["unit"], d.get("ms_per_step", 0), d.get("dtype")))
r = d.get("roofline")
if r:
    print("roofline: %s %s %.1f/%.0f %s frac %.4f avg %.2f us share %.3f traffic %s launches/step %s" % (
        r["kernel"], r["bound"], r["achieved"], r["peak"], r["unit"], r["frac"], r["avg_us"], r.get("share_of_step", 0), r.get("traffic"), r.get("launches_per_step")))
p = d.get("pipeline")
if p:
    print("pipeline:", {k: p[k] for k in p})
for k in ("single_stream", "device_only_images_per_s", "cpu_baseline", "latency_bs1"):
    if d.get(k) is not None:
        print(k + ":", json.dumps(d[k])[:400])
for name, e in (d.get("extras") or {}).items():
    if "error" in e:
        print("extra", name, "ERROR", e["error"]); continue
    v = e.get("images_per_s", e.get("value"))
    rr = e.get("roofline") or {}
    pp = e.get("pipeline") or {}
    print("extra %-28s %9.1f img/s  %.4f ms  dom %s frac %s  floor-frac %s  %s" % (name, v, e.get("ms_per_step", 0), rr.get("kernel"), rr.get("frac"), pp.get("frac_of_floor"),
                                                                               {k: e[k] for k in ("allreduce_us_per_step",) if k in e}))
for k in (d.get("kernels") or [])[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("  %-44s x%-2d %8.2f us  share %.4f  %s %8.1f %s frac %.4f" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["share"], k["bound"], k["achieved"], k["unit"], k["frac"]))
