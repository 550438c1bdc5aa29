#!/bin/bash
# the two other named inference workloads, three runs each (run-to-run spread of the autotuned configuration)
for i in 1 2 3; do
python3 bench.py --no-cpu-baseline --no-latency --steps 60 --warmup 20 2>/dev/null | python3 -c '
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = d["extras"]
print("416 bs32 %.0f   608 bs32 %.0f   0.5x 416 bs128 %.0f img/s" % (d["value"], e["infer_608_bs32"]["images_per_s"], e["infer_0.5x_416_bs128"]["images_per_s"]))'
done
