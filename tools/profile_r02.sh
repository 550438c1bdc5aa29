#!/bin/bash
# Every rocprofv3 summary committed under profiles/ for round 2 (run on the GPU box from the repo root): bash tools/profile_r02.sh
# Output: gpurun_out/prof_r02/*  (small summaries only; the raw traces are deleted)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-latency --no-extras"
stats() {   # name, bench args...
    local name=$1; shift
    rm -rf /tmp/yn_prof_$name
    rocprofv3 --kernel-trace --stats -d /tmp/yn_prof_$name -o run --output-format csv -- python3 $R/bench.py "$@" > $O/$name.log 2>&1
    cp $(find /tmp/yn_prof_$name -name "*kernel_stats.csv" | head -1) $O/r02_kernel_stats_$name.csv
}
stats bs32_416 --steps 100 --warmup 20 $B
stats bs32_416_1stream --steps 400 --warmup 20 --streams 1 --launch eager $B
python3 $R/tools/concurrency.py $(find /tmp/yn_prof_bs32_416 -name "*kernel_trace.csv" | head -1) $(find /tmp/yn_prof_bs32_416_1stream -name "*kernel_trace.csv" | head -1) > $O/r02_4stream_concurrency.md 2>$O/concurrency.err
stats train_608_bs32_f16 --train --dtype f16 --size 608 --batch 32 --steps 100 --warmup 5
stats train_608_bs32_f32 --train --dtype f32 --size 608 --batch 32 --steps 100 --warmup 5
# per-layer HBM traffic: two separate --pmc passes, kernel trace only
ARGS="--steps 3 --warmup 2 --streams 1 --launch eager --profile-steps 1 $B"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/yn_pl_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/yn_pl_$c -o run --output-format csv -- python3 $R/bench.py $ARGS --dump-layers $O/layers_$c.json > $O/pl_$c.log 2>&1
done
python3 $R/tools/pmc_layers.py $(find /tmp/yn_pl_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/yn_pl_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
    $O/layers_FETCH_SIZE.json $O/r02_pmc_layers.json > $O/r02_pmc_layers.md 2> $O/pmc_layers.err
# SQ counters (PMC serialises dispatches: these are per-kernel, not contention, numbers)
rm -rf /tmp/yn_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
    -d /tmp/yn_sq -o run --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --profile-steps 1 $B > $O/sq.log 2>&1
python3 $R/tools/sq_summary.py $(find /tmp/yn_sq -name "*counter_collection.csv" | head -1) > $O/r02_sq_counters.txt 2> $O/sq.err
ls -la $O
