#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "nms or postprocess or infer or config or tta or pack" 2>&1 | tail -3
bash tools/ab.sh "416"
python3 bench.py --no-extras --no-cpu-baseline --no-latency --steps 60 --warmup 20 --streams 1 --launch eager --layers 2>&1 >/dev/null | grep "nms\.sort" | awk '{printf "%-32s %-40s %7s\n",$1,$2,$3}'
bash tools/l608b1.sh | grep nms
bash tools/lat.sh
