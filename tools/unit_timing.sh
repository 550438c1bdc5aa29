# phase timing of the fused ShuffleV2 unit kernel (debug build with printf)
YN_EXTRA_FLAGS=-DYN_EXP_TIMING python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --no-graph 2>/dev/null | grep "^unit" | sort -k3,3 -k5,5n | awk "{print}" | sed -n "1,3p;100,102p;400,403p;800,803p"
