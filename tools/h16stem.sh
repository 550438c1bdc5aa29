#!/bin/bash
python3 -m pytest tests/test_gpu_train_h16.py -q -m gpu -x 2>&1 | tail -3
python3 bench.py --train --dtype f16 --size 608 --batch 32 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_h16stem -- python3 $GRAFT_REPO_ROOT/bench.py --train --dtype f16 --size 608 --batch 32 --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_h16stem -name "*kernel_stats.csv" | head -1); grep -E "hstem|hmaxpool" $f | cut -c1-160
