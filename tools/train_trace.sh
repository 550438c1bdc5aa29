#!/bin/bash
# kernel trace of 12 fp16 training steps -> per-launch breakdown of a kernel family:  bash tools/train_trace.sh [pattern]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/yn_tt
rocprofv3 --kernel-trace -d /tmp/yn_tt -o run --output-format csv -- python3 $R/bench.py --train --dtype f16 --size 608 --batch 32 --steps 10 --warmup 2 > /tmp/yn_tt.log 2>&1
F=$(find /tmp/yn_tt -name "*kernel_trace.csv" | head -1)
for p in "$@"; do python3 $R/tools/train_layers.py $F "$p"; done
