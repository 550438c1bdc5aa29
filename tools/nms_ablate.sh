#!/bin/bash
# wall-time ablation of the NMS kernels in the 4-stream bench (timing only: skipped kernels leave wrong outputs)
for m in 0 1 2 4 6 7; do bash tools/ab.sh "nms_skip=$m" YN_DBG_NMS_SKIP=$m; done
