#!/bin/bash
# single-stream throughput + bs=1 latencies:  bash tools/lat2.sh [ENV=VAL ...]
env "$@" python3 bench.py --no-cpu-baseline --steps 60 --warmup 20 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
l = d["latency_bs1"]
print("%.1f img/s; single stream %.1f; bs1 416 eager/graph p50 %.4f / %.4f ms; 608 %.4f / %.4f ms; 608bs32 %.0f" % (d["value"], d["single_stream"]["images_per_s"],
      l["416x416"]["eager"]["p50_ms"], l["416x416"]["hipgraph"]["p50_ms"], l["608x608"]["eager"]["p50_ms"], l["608x608"]["hipgraph"]["p50_ms"], d["extras"]["infer_608_bs32"]["images_per_s"]))
'
