#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "configuration or channel_shuffle or pointwise" 2>&1 | tail -2
bash tools/lat.sh
bash tools/lat.sh
YN_BS1=1 python3 bench.py --batch 1 --no-extras --no-cpu-baseline --no-latency --steps 200 --warmup 30 --streams 1 --launch eager --layers 2>&1 >/dev/null | awk '{printf "%-32s %-40s %7s\n",$1,$2,$3}' | grep -c "128>"
