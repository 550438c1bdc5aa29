#!/bin/bash
# Kernel stats of the training step (DTYPE=f16|f32, default f16) on the GPU box, per step.   bash tools/train_ab.sh [size] [steps] [env assignments to compare, e.g. YN_RED_G=512]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/train_ab
mkdir -p $O
S=${1:-608}; N=${2:-30}; shift; shift
cd /tmp && export TMPDIR=/tmp
for f in base "$@"; do
    [ "$f" != base ] && export "$f"
    tag=$(echo $f | tr '= ' '__')
    rm -rf /tmp/yn_tab_$tag
    rocprofv3 --kernel-trace --stats -d /tmp/yn_tab_$tag -o run --output-format csv -- python3 $R/bench.py --train --dtype ${DTYPE:-f16} --size $S --batch 32 --steps $N --warmup 5 > $O/log_$tag.txt 2>&1
    cp $(find /tmp/yn_tab_$tag -name "*kernel_stats.csv" | head -1) $O/stats_$tag.csv
    echo "== $f: $(grep -o '"ms_per_step": [0-9.]*' $O/log_$tag.txt)"
    python3 - $O/stats_$tag.csv $((N + 5)) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("  %-60s calls/step %6.1f  ms/step %6.3f  avg us %7.1f  min %6.1f max %6.1f" % (r["Name"].replace("ynk::", "")[:60], float(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
print("  total kernel ms/step %.3f" % (tot / n / 1e6))
PY
    [ "$f" != base ] && unset "${f%%=*}"
done
python3 $R/tools/train_timeline.py $(find /tmp/yn_tab_base -name "*kernel_trace.csv" | head -1) > $O/timeline_base.txt 2>&1
