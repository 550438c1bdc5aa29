#!/bin/bash
# GPU box: KC=64 split configurations — parity of every configuration, then A/B and the per-layer picks
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "configuration or channel_shuffle or pointwise or invariance or network" 2>&1 | tail -5
for i in 1 2; do bash tools/ab.sh "kc64-enabled"; done
python3 bench.py --no-extras --no-cpu-baseline --no-latency --steps 100 --warmup 20 --streams 1 --launch eager --layers 2>&1 >/dev/null | grep -E "gemm_split" | head -40
