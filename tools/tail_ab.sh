#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "head_tail or fused_head or grouped or postprocess or network or invariance" 2>&1 | tail -3
for i in 1 2; do bash tools/ab.sh "tail-fused"; bash tools/ab.sh "tail-off" YN_TAIL_FUSE=0; done
python3 bench.py --no-extras --no-cpu-baseline --no-latency --steps 60 --warmup 20 --streams 1 --launch eager --layers 2>&1 >/dev/null | grep -E "head_det" | awk '{printf "%-28s %-34s %7s\n",$1,$2,$3}'
