#!/bin/bash
# Same-box A/B of the training step between source trees:  bash tools/ab_train.sh <dtype> dirA dirB ...   ('.' = this tree); three rounds, interleaved
DT=$1; shift
for i in 1 2 3; do
for d in "$@"; do
(cd $d && python3 bench.py --train --dtype $DT --size 608 --batch 32 --steps 40 --warmup 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s' % '$d', d['ms_per_step'])")
done; done
