#!/usr/bin/env python3
"""sha1 of the raw heads and detections of one seeded call - for A/Bs of environment switches that a process reads once (YN_DWPW_PIPE,
YN_DOWN_PIPE, ...): run it under both settings and compare the lines.   python3 tools/ab_hash.py <S> <B> [backbone] [classes]"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_nano_amd import arch, capi, weights  # noqa: E402

S, B = int(sys.argv[1]), int(sys.argv[2])
backbone = sys.argv[3] if len(sys.argv) > 3 else "1.0x"
C = int(sys.argv[4]) if len(sys.argv) > 4 else 80
anchors = arch.MULTI_ANCHOR_SIZE_COCO if C == 80 else arch.MULTI_ANCHOR_SIZE
h = capi.Handle(S, C, anchors, backbone, 0.001, 0.5, max_batch=B)
h.load_state_dict(weights.make_state_dict(backbone, C))
h.fold_bn()
x = torch.as_tensor(weights.make_input(B, S, seed=9)).cuda()
m = hashlib.sha1()
for t in h.forward_raw(x):
    m.update(t.contiguous().cpu().numpy().tobytes())
out = h.infer(x)
counts = out[4].cpu().tolist()
if min(counts) < 0:                                         # yn_infer's range mark: the results are invalid, nothing to hash
    raise SystemExit("ab_hash: yn_infer flagged an activation outside the split-f16 range (negative counts)")
for b in range(B):
    for t in out[:4]:
        m.update(t[b, :counts[b]].contiguous().cpu().numpy().tobytes())
print("hash", S, B, backbone, C, sum(counts), m.hexdigest())
