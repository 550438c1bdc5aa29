#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
bash tools/lat.sh
bash tools/lat.sh
