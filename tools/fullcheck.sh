#!/bin/bash
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 3000 gpurun_out/bench_default.json
