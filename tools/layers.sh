#!/bin/bash
# Per-layer HIP-event table of one stream (GPU box, repo root):  bash tools/layers.sh [ENV=VAL ...] [-- bench args, e.g. --size 608 --batch 1]
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" == "--" ] && shift
env "${ENVS[@]}" python3 bench.py --no-cpu-baseline --no-latency --no-extras --steps 30 --warmup 10 --streams 1 --launch eager --layers "$@" 2>&1 >/dev/null | grep " us " | cut -c1-120
