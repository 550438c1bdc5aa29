#!/bin/bash
# Per-layer HBM traffic of the default bench workload: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel trace only.
# Run on the GPU box from the repo root:  bash tools/pmc_pass.sh   -> gpurun_out/pmc_layers.{json,md}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --streams 1 --launch eager --profile-steps 1 --no-cpu-baseline --no-latency"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pl_$c
    rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pl_$c -o run --output-format csv -- python3 $R/bench.py $ARGS --dump-layers $R/gpurun_out/layers_$c.json > $R/gpurun_out/pl_$c.log 2>&1
done
F=$(find $R/gpurun_out/pl_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $R/gpurun_out/pl_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_layers.py $F $W $R/gpurun_out/layers_FETCH_SIZE.json $R/gpurun_out/pmc_layers.json > $R/gpurun_out/pmc_layers.md
tail -5 $R/gpurun_out/pmc_layers.md
