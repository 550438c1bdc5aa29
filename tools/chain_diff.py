"""debug: where do the unit-chain and the three-kernel path differ?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from yolo_nano_amd import arch, capi, weights
S, B, bb, C = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], 20
h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, bb, 0.001, 0.5, max_batch=B)
h.load_state_dict(weights.make_state_dict(bb, C)); h.fold_bn()
x = torch.as_tensor(weights.make_input(B, S, seed=S + B)).cuda()
h.unit_chain(True); a = [t.clone() for t in h.forward_raw(x)]
h.unit_chain(False); b = [t.clone() for t in h.forward_raw(x)]
h.unit_chain(False); c = [t.clone() for t in h.forward_raw(x)]
for u, v, w in zip(a, b, c):
    d = (u - v).abs()
    print(tuple(u.shape), "max|d|", float(d.max()), "n!=", int((d > 0).sum()), "of", d.numel(), "rerun-equal", bool(torch.equal(v, w)), "max|v|", float(v.abs().max()))
