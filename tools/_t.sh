python3 -m pytest tests/test_gpu_parity.py -q -x -k "unit_chain or shuffle or net_ or config2 or config4 or backbone_taps or size_sweep" 2>&1 | tail -3
bash tools/ab_env.sh "--no-cpu-baseline --no-latency --no-extras --steps 200 --warmup 30" YN_CHAIN_V=1 YN_CHAIN_BAL=0 YN_CHAIN_BAL=1
python3 bench.py --no-cpu-baseline --no-latency --no-extras --steps 50 --warmup 10 --streams 1 > gpurun_out/c2.json 2>/dev/null; python3 tools/bench_summary.py gpurun_out/c2.json 6
