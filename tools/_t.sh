python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_extras.py -q -x -k "nms or postprocess or infer or tta or merge or config5 or yolonano" 2>&1 | tail -3
bash tools/ab_env.sh "--no-cpu-baseline --no-latency --no-extras --steps 200 --warmup 30" -
python3 bench.py --no-cpu-baseline --no-latency --no-extras --steps 30 --warmup 10 --streams 1 --layers 2>&1 >/dev/null | grep "nms\." | cut -c1-80
python3 bench.py --no-cpu-baseline --no-latency --no-extras --steps 30 --warmup 10 --streams 1 --layers --size 608 2>&1 >/dev/null | grep "nms\." | cut -c1-80
