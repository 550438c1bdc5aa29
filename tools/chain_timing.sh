# phase cycle counts of the unit-chain kernels (debug build with printf); cycle-counter ticks.  bash tools/chain_timing.sh [bf]
BF=${1:-116}
YN_EXTRA_FLAGS=-DYN_EXP_TIMING python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-extras --streams 1 --launch eager --profile-steps 1 2>&1 | grep "^chain.* bf $BF " | awk "NR%10==1" | tail -12
python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
