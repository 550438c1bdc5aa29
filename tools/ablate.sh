#!/bin/bash
# Marginal wall-time cost of each region of the network in the default (4-stream) run: the region's launches are dropped after the
# warm-up passes (the arena still holds their outputs, so downstream work is unchanged).  bash tools/ablate.sh
P="--no-cpu-baseline --no-latency --no-extras --steps 200 --warmup 30"
bash tools/ab_env.sh "$P" -
for r in "stem" "stage2.0" "stage2.1,stage2.2,stage2.3" "stage3.0" "stage3.1,stage3.2,stage3.3,stage3.4,stage3.5,stage3.6,stage3.7" "stage4.0" "stage4.1,stage4.2,stage4.3" "conv1x1" "smooth" "head_det_*.0,head_det_*.2" "head_det_*.1,head_det_*.3" "head_det_*.4"; do
  bash tools/ab_env.sh "$P" "YN_DBG_SKIP_LAYERS=$r" | head -1
done
bash tools/ab_env.sh "$P" - | head -1
