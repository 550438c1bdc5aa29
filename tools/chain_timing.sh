# phase cycle counts of unit_chain_kernel (debug build with printf); s_memtime/cycle counter ticks
YN_EXTRA_FLAGS=-DYN_EXP_TIMING python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --launch eager --profile-steps 0 2>/dev/null | grep "^chain" | sort -k3,3n -k5,5n | awk '{print}' | sed -n "1,4p;60,63p;200,203p;300,303p;420,423p"
python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
