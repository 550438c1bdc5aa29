#!/bin/bash
# A/B of environment switches on ONE box:  bash tools/ab_env.sh "<bench args>" "VAR=1 VAR2=x" "VAR=2" ...   ("-" = no variables); two rounds
ARGS=$1; shift
for i in 1 2; do
for e in "$@"; do
[ "$e" == "-" ] && ee="" || ee="$e"
env $ee python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); det=json.load(open(d['detail_file'])) if d.get('detail_file') else d; print('%-28s' % '$e', d['value'], d['ms_per_step'], 'dev_only', det.get('device_only_images_per_s'), 'sumk', (det.get('pipeline') or {}).get('sum_kernel_ms'))"
done; done
