#!/bin/bash
# A/B helper (GPU box, repo root):  bash tools/ab.sh "label" ENV=VAL ... -- extra bench args
# prints: label  images/s  ms/step  sum of kernel ms (one stream, HIP events)
LABEL=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" == "--" ] && shift
env "${ENVS[@]}" python3 bench.py --no-extras --no-cpu-baseline --no-latency --steps 200 --warmup 30 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %9.1f img/s  %.4f ms/step  sum_kernel_ms %.4f' % ('$LABEL', d['value'], d['ms_per_step'], (d.get('pipeline') or {}).get('sum_kernel_ms', 0)))
"
