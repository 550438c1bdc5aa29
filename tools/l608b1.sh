#!/bin/bash
python3 bench.py --size 608 --batch 1 --no-extras --no-cpu-baseline --no-latency --steps 200 --warmup 30 --streams 1 --launch eager --layers 2>&1 >/dev/null | awk '{printf "%-32s %-40s %7s\n",$1,$2,$3}' | tail -16
