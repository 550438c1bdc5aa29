#!/usr/bin/env python3
"""Per-phase cycle sums of stage_pipe_kernel (a build with YN_EXTRA_FLAGS=-DYN_EXP_STAGE_TIMING prints them): python3 tools/stage_timing.py <publish_early 0|1> [S] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_nano_amd import arch, capi, weights  # noqa: E402

early = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 416
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    h = capi.Handle(S, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=B, stream=st)
    h.load_state_dict(weights.make_state_dict("1.0x", 80))
    h.fold_bn()
    x = torch.as_tensor(weights.make_input(B, S, seed=7)).cuda()
    h.stage_fuse(1, bool(early))
    for _ in range(3):
        h.forward_raw(x)
    st.synchronize()
