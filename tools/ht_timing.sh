# phase cycle counts of head_tail_group_kernel (debug build with printf)
YN_EXTRA_FLAGS=-DYN_EXP_TIMING python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-extras --streams 1 --launch eager --profile-steps 1 2>&1 | grep "^headtail" | awk 'NR%5==1' | tail -10
python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
