#!/bin/bash
# GPU box, repo root: the whole -m gpu suite (no -x, full log), smoke(), then the bench in the driver's form.
#   bash tools/gpucheck.sh [tag] [pytest -k expression]
TAG=${1:-check}; K=${2:-}
mkdir -p gpurun_out
if [ -n "$K" ]; then python3 -m pytest tests -q -m gpu -k "$K" > gpurun_out/${TAG}_pytest.log 2>&1
else python3 -m pytest tests -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; fi
tail -n 30 gpurun_out/${TAG}_pytest.log | cut -c1-400
if [ -z "$K" ]; then
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cp bench_detail.json gpurun_out/${TAG}_bench_detail.json
wc -c gpurun_out/${TAG}_bench.json; python3 tools/bench_summary.py gpurun_out/${TAG}_bench.json
fi
