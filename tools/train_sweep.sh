cd /tmp; export TMPDIR=/tmp
for cfg in "512 2048" "256 2048" "1024 2048"; do
  set -- $cfg
  export YN_WG_SLICES=$1 YN_STEM_G=$2
  d=$GRAFT_REPO_ROOT/gpurun_out/sz_$1_$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --train --size 608 --batch 32 --steps 5 --warmup 2 > $d.log 2>&1
  echo "== WG_SLICES=$1 STEM_G=$2"
  python3 - <<PY
import csv,glob
f=glob.glob("$d/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("step ms", tot/1e6/7)
for r in rows:
    if 'wgrad' in r['Name']: print("  %-60s calls %4s avg %8.1f us"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
