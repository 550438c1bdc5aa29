python3 -m pytest tests/test_gpu_parity.py -q -x -k "unit_chain or shuffle or net_ or config2 or config4 or backbone_taps or size_sweep" 2>&1 | tail -3
bash tools/ab_env.sh "--no-cpu-baseline --no-latency --no-extras --steps 200 --warmup 30" - YN_CHAIN_MIN4=4096
bash tools/ab_env.sh "--no-cpu-baseline --no-latency --no-extras --steps 100 --warmup 20 --streams 1" - YN_CHAIN_MIN4=4096
YN_CHAIN_MIN4=4096 python3 bench.py --no-cpu-baseline --no-latency --no-extras --steps 30 --warmup 10 --streams 1 --layers 2>&1 >/dev/null | grep "stage4"
