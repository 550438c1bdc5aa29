#!/bin/bash
# The compiler's per-kernel resource report for every kernel of the library (no GPU needed):  bash tools/resource_table.sh > profiles/r06_resource_usage.txt
# (Function Name / VGPRs / AGPRs / SGPRs / ScratchSize / Occupancy / static LDS per kernel; tools/concurrency.py and tools/resource_usage.py read it.)
R=$(cd "$(dirname "$0")/.." && pwd)
for f in kernels_conv kernels_post kernels_train kernels_bwd kernels_h16 kernels_chain kernels_pipe kernels_stage; do
    extra=""; [ $f == kernels_post ] && extra="-ffp-contract=off"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -Wno-pass-failed $extra --cuda-device-only -c $R/yolo-nano_amd/csrc/$f.hip -o /dev/null \
        -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|TotalSGPRs|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/^.*remark: */remark: /; s/ \[-Rpass-analysis=kernel-resource-usage\]//'
done
