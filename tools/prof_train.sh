#!/bin/bash
# Kernel-trace summary of the training step (run on the GPU box from the repo root):  bash tools/prof_train.sh f16|f32 [steps]
# -> gpurun_out/prof_train_<dtype>/ ... _kernel_stats.csv   (warm-up 5 steps: the autotune launches of the fp32 step stay a small share at >= 100 steps)
DT=${1:-f16}; STEPS=${2:-100}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train_$DT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train_$DT -o run --output-format csv -- python3 $R/bench.py --train --dtype $DT --size 608 --batch 32 --steps $STEPS --warmup 5 > $R/gpurun_out/prof_train_$DT.log 2>&1
F=$(find $R/gpurun_out/prof_train_$DT -name "*kernel_stats.csv" | head -1)
head -40 $F | cut -c1-200
