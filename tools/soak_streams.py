#!/usr/bin/env python3
"""Run-to-run determinism under contention: four handles on four streams (bench.py's default shape), every handle infers ITS input again and
again while the others run; the detections of every call must equal the handle's first call bit for bit.  A race between an LDS-DMA piece and
its reader (unit_pipe_kernel, pw_pipe_kernel, head_tail_group_kernel), or a missing barrier, would show up as a changing hash under load long
before it shows up alone.   python3 tools/soak_streams.py [calls per stream = 300] [S = 416] [B = 32]"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolo_nano_amd import arch, capi, weights  # noqa: E402


def digest(out, B):
    counts = out[4].cpu().tolist()
    if min(counts) < 0:
        raise SystemExit("soak_streams: range flag set (negative counts)")
    m = hashlib.sha1()
    for b in range(B):
        for t in out[:4]:
            m.update(t[b, :counts[b]].contiguous().cpu().numpy().tobytes())
    return sum(counts), m.hexdigest()


def main(calls=300, S=416, B=32, ns=4):
    sd = weights.make_state_dict("1.0x", 80)
    streams = [torch.cuda.Stream() for _ in range(ns)]
    hs, xs, outs = [], [], []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            h = capi.Handle(S, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=B, stream=st)
            h.load_state_dict(sd)
            h.fold_bn()
            hs.append(h)
            xs.append(torch.as_tensor(weights.make_input(B, S, seed=40 + k)).cuda())
            outs.append([h.alloc_outputs(B) for _ in range(2)])
    ref = []
    for k, st in enumerate(streams):                        # reference: each handle alone
        with torch.cuda.stream(st):
            hs[k].infer(xs[k], outs[k][0])
        st.synchronize()
        ref.append(digest(outs[k][0], B))
    bad = 0
    for i in range(calls):
        for k, st in enumerate(streams):                    # all four in flight
            with torch.cuda.stream(st):
                hs[k].infer(xs[k], outs[k][i & 1])
        if i % 10 == 9 or i == calls - 1:                   # (hashing synchronises: every tenth round)
            for k, st in enumerate(streams):
                st.synchronize()
                if digest(outs[k][i & 1], B) != ref[k]:
                    bad += 1
                    print("MISMATCH round", i, "stream", k)
    for h in hs:
        h.close()
    print("soak_streams: %d rounds x %d streams, S=%d B=%d, kept %s: %s" % (calls, ns, S, B, [r[0] for r in ref], "all equal" if not bad else "%d MISMATCHES" % bad))
    return bad


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    sys.exit(1 if main(*a) else 0)
