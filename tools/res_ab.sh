#!/bin/bash
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "nms or postprocess or infer or config or tta or pack" 2>&1 | tail -3
bash tools/l608b1.sh | grep nms
bash tools/lat.sh
bash tools/lat.sh
