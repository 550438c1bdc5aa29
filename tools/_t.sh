python3 -m pytest tests/test_gpu_parity.py -q -x -k "decode or infer or head_tail or grouped or yolonano" 2>&1 | tail -3
bash tools/ab_env.sh "--no-cpu-baseline --no-latency --no-extras --steps 200 --warmup 30" -
bash tools/ht_timing.sh | tail -3
