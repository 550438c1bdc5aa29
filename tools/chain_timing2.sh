# dw-phase detail of the split chain kernel (debug build with printf)
YN_EXTRA_FLAGS=-DYN_EXP_TIMING python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-extras --streams 1 --launch eager --profile-steps 0 2>/dev/null | grep "^chaindw bf 116" | awk "NR%40==1" | tail -12
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-extras --streams 1 --launch eager --profile-steps 0 2>/dev/null | grep "^chains bf 116" | awk "NR%40==1" | tail -6
python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
