P="--no-cpu-baseline --no-latency --no-extras --steps 200 --warmup 30"
for i in 1 2; do
(cd _ab_old && python3 bench.py $P 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLD', d['value'], d['ms_per_step'], d['device_only_images_per_s'])")
python3 bench.py $P 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW', d['value'], d['ms_per_step'], d['device_only_images_per_s'])"
done
(cd _ab_old && python3 bench.py --train --size 608 --batch 32 --steps 12 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLD train', d['value'], d['ms_per_step'])")
python3 bench.py --train --size 608 --batch 32 --steps 12 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW train', d['value'], d['ms_per_step'])"
rocm-smi --showclocks 2>/dev/null | head -20
