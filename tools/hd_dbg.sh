#!/bin/bash
# phase ablation of head_decode_kernel (debug): YN_HD_DBG bit0 = skip the GEMM, bit1 = skip the decode phase (single launches: YN_GROUP=0)
for d in 0 1 2 3; do
  echo "dbg=$d"
  YN_GROUP=0 YN_HD_DBG=$d python3 bench.py --no-extras --no-cpu-baseline --no-latency --steps 30 --warmup 10 --streams 1 --launch eager --layers 2>&1 >/dev/null | grep -E "\+decode" | awk '{printf "%-28s %-34s %7s\n",$1,$2,$3}'
done
