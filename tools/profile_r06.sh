#!/bin/bash
# Every rocprofv3 summary committed under profiles/ for round 5 (run on the GPU box from the repo root): bash tools/profile_r06.sh [part ...]
#   parts: stats pmc sq train   (default: all)       Output: gpurun_out/prof_r06/*  (small summaries only; raw traces stay in /tmp)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r06
mkdir -p $O
PARTS=${@:-stats pmc sq train}
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-latency --no-extras"
stats() {   # name, bench args...
    local name=$1; shift
    rm -rf /tmp/yn_prof_$name
    rocprofv3 --kernel-trace --stats -d /tmp/yn_prof_$name -o run --output-format csv -- python3 $R/bench.py "$@" > $O/$name.log 2>&1
    cp $(find /tmp/yn_prof_$name -name "*kernel_stats.csv" | head -1) $O/r06_kernel_stats_$name.csv
}
pmc_layers() {   # name, bench args selecting the workload
    local name=$1; shift
    local ARGS="--steps 3 --warmup 2 --streams 1 --launch eager --profile-steps 1 $B $@"
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/yn_pl_${name}_$c
        rocprofv3 --kernel-trace --pmc $c -d /tmp/yn_pl_${name}_$c -o run --output-format csv -- python3 $R/bench.py $ARGS --dump-layers $O/layers_${name}_$c.json > $O/pl_${name}_$c.log 2>&1
    done
    python3 $R/tools/pmc_layers.py $(find /tmp/yn_pl_${name}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/yn_pl_${name}_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
        $O/layers_${name}_FETCH_SIZE.json $O/r06_pmc_layers$2.json > $O/r06_pmc_layers$2.md 2> $O/pmc_layers_${name}.err
}
for part in $PARTS; do
case $part in
stats)
    stats bs32_416 --steps 100 --warmup 20 $B
    stats bs32_416_1stream --steps 400 --warmup 20 --streams 1 --launch eager $B
    # per-workgroup footprints: the compiler's report (committed: tools/resource_table.sh) + the dynamic LDS of every launch shape of the default run
    YN_LOG_LDS=1 python3 $R/bench.py --steps 4 --warmup 2 $B 2>&1 >/dev/null | grep "^yn_lds" | sort -u > $O/r06_dynamic_lds.txt
    python3 $R/tools/concurrency.py $(find /tmp/yn_prof_bs32_416 -name "*kernel_trace.csv" | head -1) $(find /tmp/yn_prof_bs32_416_1stream -name "*kernel_trace.csv" | head -1) \
        $R/profiles/r06_resource_usage.txt $O/r06_dynamic_lds.txt > $O/r06_4stream_concurrency.md 2>$O/concurrency.err
    stats 608_bs32_1stream --size 608 --steps 200 --warmup 20 --streams 1 --launch eager $B
    stats 05x_bs128_1stream --backbone 0.5x --batch 128 --steps 200 --warmup 20 --streams 1 --launch eager $B
    ;;
train)
    # the autotuner's timing launches must not be in the steady-state table: an un-profiled run of each step first writes the tune table
    # (YN_TUNE_FILE: bench.py loads it at start when it exists and saves it at exit), the profiled run adopts it and times nothing
    for dt in f16 f32; do
        rm -f /tmp/yn_tune_train_$dt.txt
        YN_TUNE_FILE=/tmp/yn_tune_train_$dt.txt python3 $R/bench.py --train --dtype $dt --size 608 --batch 32 --steps 6 --warmup 4 > $O/train_warm_$dt.log 2>&1
        YN_TUNE_FILE=/tmp/yn_tune_train_$dt.txt stats train_608_bs32_$dt --train --dtype $dt --size 608 --batch 32 --steps 100 --warmup 8
        # one step of the trace, per queue: busy / idle (why sum(kernel time) != wall step on a multi-stream executor)
        python3 $R/tools/train_timeline.py $(find /tmp/yn_prof_train_608_bs32_$dt -name "*kernel_trace.csv" | head -1) | tail -8 > $O/r06_train_timeline_$dt.txt 2>&1
    done
    ;;
pmc)
    # per-layer HBM traffic: two separate --pmc passes each, kernel trace only
    name=416; ARGS="--steps 3 --warmup 2 --streams 1 --launch eager --profile-steps 1 $B"
    for w in "416::" "608:_608_bs32:--size 608" "05x:_05x_bs128:--backbone 0.5x --batch 128"; do
        name=${w%%:*}; rest=${w#*:}; suf=${rest%%:*}; extra=${rest#*:}
        for c in FETCH_SIZE WRITE_SIZE; do
            rm -rf /tmp/yn_pl_${name}_$c
            rocprofv3 --kernel-trace --pmc $c -d /tmp/yn_pl_${name}_$c -o run --output-format csv -- python3 $R/bench.py $ARGS $extra --dump-layers $O/layers_${name}_$c.json > $O/pl_${name}_$c.log 2>&1
        done
        python3 $R/tools/pmc_layers.py $(find /tmp/yn_pl_${name}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/yn_pl_${name}_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
            $O/layers_${name}_FETCH_SIZE.json $O/r06_pmc_layers$suf.json > $O/r06_pmc_layers$suf.md 2> $O/pmc_layers_$name.err
    done
    ;;
sq)
    # SQ counters (PMC serialises dispatches: these are per-kernel, not contention, numbers)
    rm -rf /tmp/yn_sq
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
        -d /tmp/yn_sq -o run --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --profile-steps 1 $B > $O/sq.log 2>&1
    python3 $R/tools/sq_summary.py $(find /tmp/yn_sq -name "*counter_collection.csv" | head -1) > $O/r06_sq_counters.txt 2> $O/sq.err
    ;;
esac
done
ls -la $O | head -40
