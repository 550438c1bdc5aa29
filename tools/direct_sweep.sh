# per-layer time of a few GEMM layers under every register-direct configuration (indices 36..)
for c in "$@"; do
    echo "== cfg $c"
    YN_PW_FORCE_CFG=$c python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --launch eager --layers 2>&1 >/dev/null | grep -E "stage4.2.b2.pw|stage3.2.b2.pw|head_det_1.1|conv1x1_0|head_det_2.1" | awk '{printf "%-28s %-30s %7.1f us\n",$1,$2,$3}'
done
