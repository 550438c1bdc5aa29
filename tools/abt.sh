#!/bin/bash
# A/B helper for the training step:  bash tools/abt.sh "label" ENV=VAL ... -- extra bench args
LABEL=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" == "--" ] && shift
env "${ENVS[@]}" python3 bench.py --train --dtype f16 --size 608 --batch 32 --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %9.1f img/s  %.4f ms/step  %s finite=%s' % ('$LABEL', d['value'], d['ms_per_step'], d['dtype'], d['finite']))
"
