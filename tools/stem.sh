#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dense3x3 or stem or maxpool or network or invariance or heads or tap" 2>&1 | tail -3
for i in 1 2; do bash tools/ab.sh "stem-sgpr"; done
python3 bench.py --no-extras --no-cpu-baseline --no-latency --steps 100 --warmup 20 --streams 1 --launch eager --layers 2>&1 >/dev/null | grep -E "stem" | head
