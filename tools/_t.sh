python3 -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
bash tools/ab_env.sh "--no-cpu-baseline --no-latency --no-extras --steps 200 --warmup 30" -
for S in 416 608; do for g in "" "--graph"; do
python3 bench.py --size $S --batch 1 --latency 1000 --no-cpu-baseline $g 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$S $g', d['value'], d['p99_ms'])"
done; done
