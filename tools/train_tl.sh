R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -f /tmp/yn_tune_train_f16.txt
YN_TUNE_FILE=/tmp/yn_tune_train_f16.txt python3 $R/bench.py --train --dtype f16 --size 608 --batch 32 --steps 6 --warmup 4 > $O/train_warm_f16.log 2>&1
rm -rf /tmp/yn_prof_tl
YN_TUNE_FILE=/tmp/yn_tune_train_f16.txt rocprofv3 --kernel-trace --stats -d /tmp/yn_prof_tl -o run --output-format csv -- python3 $R/bench.py --train --dtype f16 --size 608 --batch 32 --steps 30 --warmup 8 > $O/train_tl.log 2>&1
python3 $R/tools/train_timeline.py $(find /tmp/yn_prof_tl -name "*kernel_trace.csv" | head -1) > $O/train_timeline_f16_full.txt 2>&1
tail -6 $O/train_timeline_f16_full.txt
