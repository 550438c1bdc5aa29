#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2; do bash tools/ab.sh "hd-rowinfo"; done
python3 bench.py --no-extras --no-cpu-baseline --no-latency --steps 60 --warmup 20 --streams 1 --launch eager --layers 2>&1 >/dev/null | grep -E "decode" | awk '{printf "%-28s %-34s %7s\n",$1,$2,$3}'
