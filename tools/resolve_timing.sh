#!/bin/bash
# phase cycle counters of resolve_kernel (debug build with printf): bash tools/resolve_timing.sh
YN_EXTRA_FLAGS=-DYN_EXP_TIMING python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-extras --streams 1 --no-graph 2>/dev/null | grep "^resolve" | sort -u | tail -6
