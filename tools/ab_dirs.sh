#!/bin/bash
# A/B of whole source trees on ONE box:  bash tools/ab_dirs.sh "<bench args>" dirA dirB ...   ('.' = this tree); two rounds, interleaved
ARGS=$1; shift
for i in 1 2; do
for d in "$@"; do
(cd $d && python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s' % '$d', d['value'], d['ms_per_step'], 'dev_only', d.get('device_only_images_per_s'), 'single', (d.get('single_stream') or {}).get('images_per_s'))")
done; done
