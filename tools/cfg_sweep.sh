# time of selected GEMM layers under every pointwise tile configuration (YN_PW_FORCE_CFG), one line per configuration
for c in $(seq 0 43); do
    YN_PW_FORCE_CFG=$c python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-latency --streams 1 --launch eager --layers 2>&1 >/dev/null | grep -E "^(backbone.stage3.0.b2.pw1|backbone.stage2.0.b2.pw1|head_det_1.4|conv1x1_0|head_det_1.1|backbone.stage4.1.b2.pw1) " | awk -v c=$c '{printf "%s %s %s %.1f | ", (NR==1?c:""), (NR==1?$2:""), $1, $3} END{print ""}'
done
