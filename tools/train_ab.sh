#!/bin/bash
# A/B of the fp16 training step's BatchNorm forms on the GPU box: kernel stats of both, per step.   bash tools/train_ab.sh [size] [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/train_ab
mkdir -p $O
S=${1:-608}; N=${2:-30}
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
    export YN_BN_FUSED=$f
    rm -rf /tmp/yn_tab_$f
    rocprofv3 --kernel-trace --stats -d /tmp/yn_tab_$f -o run --output-format csv -- python3 $R/bench.py --train --dtype f16 --size $S --batch 32 --steps $N --warmup 5 > $O/log_$f.txt 2>&1
    cp $(find /tmp/yn_tab_$f -name "*kernel_stats.csv" | head -1) $O/stats_$f.csv
    echo "== YN_BN_FUSED=$f: $(grep -o '"ms_per_step": [0-9.]*' $O/log_$f.txt)"
    python3 - $O/stats_$f.csv $((N + 5)) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"])
for r in rows[:14]:
    print("  %-60s calls/step %6.1f  ms/step %6.3f  avg us %7.1f" % (r["Name"].replace("ynk::", "")[:60], float(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6, float(r["AverageNs"]) / 1e3))
print("  total kernel ms/step %.3f" % (tot / n / 1e6))
PY
done
