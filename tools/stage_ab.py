#!/usr/bin/env python3
"""stage_pipe_kernel A/B on one stream: HIP-event time of the stride-1 units of every stage, per form.
   python3 tools/stage_ab.py [S = 416] [B = 32] [backbone = 1.0x] [reps = 30]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_nano_amd import arch, capi, weights  # noqa: E402


def main(S=416, B=32, bb="1.0x", reps=30):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        h = capi.Handle(S, 80, arch.MULTI_ANCHOR_SIZE_COCO, bb, 0.001, 0.5, max_batch=B, stream=st)
        h.load_state_dict(weights.make_state_dict(bb, 80))
        h.fold_bn()
        x = torch.as_tensor(weights.make_input(B, S, seed=7)).cuda()
        ref = None
        for label, fuse, early in (("per-unit", 0, True), ("stage/early", 1, True), ("stage/deferred", 1, False), ("per-unit", 0, True), ("stage/early", 1, True)):
            h.stage_fuse(fuse, early)
            for _ in range(5):
                out = [t.clone() for t in h.forward_raw(x)]
            if ref is None:
                ref = out
            same = all(torch.equal(u, v) for u, v in zip(out, ref))
            acc = {}
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                h.forward_raw(x)
            e1.record(st)
            e1.synchronize()
            wall = e0.elapsed_time(e1) / reps * 1e3
            for _ in range(reps):
                h.profile_enable(True)
                h.forward_raw(x)
                recs = h.profile_records()
                h.profile_enable(False)
                for name, kern, ms, fl, by in recs:
                    if ".dw+pw2" in name:
                        key = (name.split(".")[1], kern)
                        acc[key] = acc.get(key, 0.0) + ms * 1e3
            tot = {}
            for (stage, kern), us in sorted(acc.items()):
                tot[stage] = tot.get(stage, 0.0) + us / reps
            print("%-15s same=%s net %.1f us | %s | %s" % (label, same, wall, "  ".join("%s %.1f" % kv for kv in sorted(tot.items())),
                                                       "  ".join("%s %.1f" % (k[1][:28], v / reps) for k, v in sorted(acc.items()) if k[0] == "stage3")))
        h.close()


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 416, int(a[1]) if len(a) > 1 else 32, a[2] if len(a) > 2 else "1.0x", int(a[3]) if len(a) > 3 else 30)
