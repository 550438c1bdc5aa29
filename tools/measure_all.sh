# every number quoted in DESIGN.md §4, one JSON line each (run on the GPU box from the repo root)
P="python3 bench.py --no-cpu-baseline"
j() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['unit'], 'ms/step', d.get('ms_per_step'), 'p99', d.get('p99_ms'), 'pcie', d.get('pcie_inclusive_images_per_s'))"; }
$P --steps 150 --warmup 30 2>/dev/null | j "416 bs32 3 streams (default)"
$P --steps 150 --warmup 30 --streams 1 2>/dev/null | j "416 bs32 1 stream"
$P --steps 150 --warmup 30 --streams 1 --conf 0.1 --nms 0.45 2>/dev/null | j "416 bs32 conf0.1 nms0.45 1 stream"
$P --steps 150 --warmup 30 --conf 0.1 --nms 0.45 2>/dev/null | j "416 bs32 conf0.1 nms0.45 3 streams"
$P --steps 60 --warmup 15 --backbone 0.5x --batch 128 2>/dev/null | j "0.5x 416 bs128"
$P --steps 80 --warmup 20 --size 608 2>/dev/null | j "608 bs32"
$P --size 608 --batch 1 --latency 1000 --graph 2>/dev/null | j "608 bs1 latency graph"
$P --size 608 --batch 1 --latency 1000 2>/dev/null | j "608 bs1 latency eager"
$P --size 416 --batch 1 --latency 1000 2>/dev/null | j "416 bs1 latency eager"
python3 bench.py --train --size 608 --batch 32 --steps 20 --warmup 5 2>/dev/null | j "train 608 bs32"
python3 bench.py --train --size 416 --batch 32 --steps 20 --warmup 5 2>/dev/null | j "train 416 bs32"
