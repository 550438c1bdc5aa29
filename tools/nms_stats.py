"""Candidate / kept-box statistics of the benchmark workload (random weights, conf 0.001, nms 0.5): python tools/nms_stats.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_nano_amd import arch, capi, weights

B, S, C = 32, 416, int(os.environ.get("NMS_STATS_C", "80"))
h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE_COCO if C == 80 else arch.MULTI_ANCHOR_SIZE, "1.0x", 0.001, 0.5, max_batch=B)
h.load_state_dict(weights.make_state_dict("1.0x", C))
h.fold_bn()
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
x = torch.randn((B, 3, S, S), generator=gen, device="cuda")
out = h.infer(x)
counts = out[4].cpu().numpy()
if counts.min() < 0:                                        # yn_infer's range mark: the results are invalid
    raise SystemExit("nms_stats: yn_infer flagged an activation outside the split-f16 range (negative counts)")
print("kept per image: min %d median %d max %d total %d" % (counts.min(), np.median(counts), counts.max(), counts.sum()))
heads = h.forward_raw(x)
bbox, cls = h.score_full(heads)
sc, ci = cls.max(-1)
for b in (0, 1):
    ok = sc[b] >= 0.001
    hist = torch.bincount(ci[b][ok], minlength=C).cpu().numpy()
    print("image %d: candidates above threshold %d of %d; per-class count: max %d, classes with >1024: %d, >256: %d, nonempty %d"
          % (b, int(ok.sum()), sc.shape[1], hist.max(), (hist > 1024).sum(), (hist > 256).sum(), (hist > 0).sum()))
    srt = np.sort(hist)[::-1]
    T = (srt + 63) // 64
    print("   largest segments:", srt[:12].tolist(), " tiles total", int((T * (T + 1) // 2).sum()), " largest segment's tiles", int(T[0] * (T[0] + 1) // 2))
