# ablation of gemm_direct_kernel on the GPU box: full / no MFMA (loads + one FMA) / no loads in the loop (first D groups only)
for v in "" "-DYN_EXP_NO_MFMA" "-DYN_EXP_NO_LOAD"; do
    YN_EXTRA_FLAGS="$v" python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
    echo "== variant '$v'"
    YN_PW_FORCE_CFG=${1:-37} python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 --launch eager --layers 2>&1 >/dev/null | grep -E "stage4.2|stage3.2.b2.pw|head_det_1.1|conv1x1_0" | awk '{printf "%-28s %-30s %7.1f us\n",$1,$2,$3}'
done
python3 -c "from yolo_nano_amd import build; build.build(force=True)" > /dev/null 2>&1
