#!/bin/bash
# headline + single-stream + bs=1 latency block of the default bench, compact:  bash tools/lat.sh [ENV=VAL ...]
env "$@" python3 bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
l = d["latency_bs1"]
print("%.1f img/s; bs1 416 eager/graph p50 %.4f / %.4f ms; 608 %.4f / %.4f ms" % (d["value"],
      l["416x416"]["eager"]["p50_ms"], l["416x416"]["hipgraph"]["p50_ms"], l["608x608"]["eager"]["p50_ms"], l["608x608"]["hipgraph"]["p50_ms"]))
'
