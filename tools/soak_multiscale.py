#!/usr/bin/env python3
"""Soak of the multi-scale training walk (tests/test_gpu_train.py::_multi_scale_run): N runs per precision, per run the worst per-tensor
ratio err / bar over all steps, the tensors over the bar and the lowest cosine.   python3 tools/soak_multiscale.py [runs] > gpurun_out/...txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T                                     # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8          # (a run takes 20-70 s of host time for its float64 oracles: 20 + 20 runs overran a 40-minute box)
g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"), allow_pickle=True)
for precision in ("f32", "f16"):
    k_mul, k_abs = T.MS_BAR[precision]
    bar = (lambda y: max(k_mul * y, k_abs)) if precision == "f32" else (lambda y: k_mul * y + k_abs)
    worst_all = 0.0
    for r in range(runs):
        worst, over, cmin, emax = 0.0, [], 1.0, 0.0
        for st in T._multi_scale_run(g, precision, T.MS_SIZES, T.MS_B):
            for n, e in st["errs"].items():
                q = e / bar(st["ey"][n])
                worst = max(worst, q); emax = max(emax, e)
                if q > 1.0:
                    over.append((st["phase"], st["it"], n, round(e, 4), round(st["ey"][n], 4)))
            cmin = min(cmin, st["cos"])
        worst_all = max(worst_all, worst)
        print("%s run %2d: worst err/bar %.3f  worst rel err %.4f  min cosine %.4f  over the bar: %s" % (precision, r, worst, emax, cmin, over[:6]), flush=True)
    print("%s: %d runs, sizes %s B=%d, bar (%g x yardstick, %g): worst err/bar %.3f" % (precision, runs, T.MS_SIZES, T.MS_B, k_mul, k_abs, worst_all), flush=True)
