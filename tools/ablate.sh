#!/bin/bash
# Marginal wall-time cost of each region of the network in the default (4-stream) run: the region's launches are dropped after the
# warm-up passes (the arena still holds their outputs, so downstream work is unchanged).  bash tools/ablate.sh
P="--no-cpu-baseline --no-latency --no-extras --steps 200 --warmup 30"
bash tools/ab_env.sh "$P" - | head -1
for r in "stem" "stage2.0" "stage2.1,stage2.2,stage2.3" "stage3.0" "stage3.1,stage3.2,stage3.3,stage3.4,stage3.5,stage3.6,stage3.7" "stage4.0" "stage4.1,stage4.2,stage4.3" "conv1x1" "smooth_1" "smooth_0,smooth_2,smooth_3" "head_det_*.0" "head_det_*.2"; do
  bash tools/ab_env.sh "$P" "YN_DBG_SKIP_LAYERS=$r" | head -1
done
for n in 1 2 4 7; do bash tools/ab_env.sh "$P" "YN_DBG_NMS_SKIP=$n" | head -1; done      # NMS: bit 0 sort, bit 1 matrix, bit 2 resolve
bash tools/ab_env.sh "$P" - | head -1
