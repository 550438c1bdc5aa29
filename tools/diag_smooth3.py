#!/usr/bin/env python3
"""Diagnostic: per-parameter relative error of the fp32 training step's gradients against the float64 oracle for the stride-32 level's layers,
over batch sizes / seeds (GPU box):  python3 tools/diag_smooth3.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T
from yolo_nano_amd import arch, weights
from oracle.torch_port import TrainNet
g = np.load(os.path.join(ROOT, "tests", "golden", "train.npz"), allow_pickle=True)
rel = lambda a, e: float(np.linalg.norm((a - e).ravel()) / max(np.linalg.norm(e.ravel()), 1e-30))
for S, B, xs, ts in ((128, 4, 60, 17), (128, 4, 61, 18), (128, 4, 21, 5), (128, 2, 60, 17), (128, 3, 60, 17), (128, 5, 60, 17), (128, 8, 60, 17), (160, 4, 60, 17)):
    h, sd = T._handle(S, 20, B, float(g["init_bias_value"]))
    x = weights.make_input(B, S, seed=xs); target = T._targets(S, 20, B, seed=ts)
    h.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(target).cuda(), lr=1e-4, update=False)
    _, g64 = TrainNet(sd, "1.0x", 20, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64).train_step(x, target, S, lr=1e-4)
    _, g32 = TrainNet(sd, "1.0x", 20, anchors=arch.MULTI_ANCHOR_SIZE).train_step(x, target, S, lr=1e-4)
    out = []
    for n in ("smooth_3.convs.0.weight", "smooth_3.convs.1.weight", "smooth_3.convs.1.bias", "smooth_2.convs.0.weight", "conv1x1_2.convs.0.weight",
              "head_det_3.0.convs.0.weight", "head_det_3.4.weight", "backbone.stage4.3.branch2.5.weight"):
        e = g64[n].numpy()
        out.append("%s %.2e/%.2e" % (n.split(".convs")[0][-12:] + n[-8:], rel(T._grad(h, n, e.shape).astype(np.float64), e), rel(g32[n].double().numpy(), e)))
    print("S %d B %d seeds %d/%d: " % (S, B, xs, ts) + "  ".join(out), flush=True)
    h.close()
