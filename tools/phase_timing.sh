#!/bin/bash
# Phase cycle counters of one kernel (GPU box, repo root): rebuilds with -DYN_EXP_TIMING (device printf), runs two steps, prints the
# sampled lines whose first word is <prefix>, rebuilds the release library.
#   bash tools/phase_timing.sh chain2 [filter] [bench args...]    unit_chain2_kernel (e.g. filter " bf 116 ")
#   bash tools/phase_timing.sh downunit | headtail | c3split | resolve
PFX=$1; FILTER=${2:-}; shift; shift
YN_EXTRA_FLAGS=-DYN_EXP_TIMING python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-extras --streams 1 --launch eager --profile-steps 1 "$@" 2>&1 | grep "^$PFX" | grep -- "$FILTER" | awk 'NR%7==1' | tail -12
python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
