import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from yolo_nano_amd import arch, capi, weights
"""Largest class segments per image of the benchmark's synthetic workload, and how many the sweep takes:  S=608 BB=0.5x B=128 python3 tools/seg_sizes.py"""
B, S, C = int(os.environ.get("B", "32")), int(os.environ.get("S", "416")), 80
BB = os.environ.get("BB", "1.0x")
h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE_COCO, BB, 0.001, 0.5, max_batch=B)
h.load_state_dict(weights.make_state_dict(BB, C)); h.fold_bn()
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
x = torch.randn((B, 3, S, S), generator=gen, device="cuda")
heads = h.forward_raw(x)
bbox, cls = h.score_full(heads)
sc, ci = cls.max(-1)
for b in range(min(B, 6)):
    ok = sc[b] >= 0.001
    hist = torch.bincount(ci[b][ok], minlength=C).cpu().numpy()
    print(b, "largest classes", np.sort(hist)[::-1][:8].tolist(), "ids", np.argsort(hist)[::-1][:4].tolist())
out = h.infer(x)
print("segments on the sweep:", h.nms_sweep_segments(B, C), "of", B, "images")
import numpy as np
bb = bbox[0].cpu().numpy(); cc = ci[0].cpu().numpy(); ss = sc[0].cpu().numpy()
for c in np.argsort(np.bincount(cc[ss >= 0.001], minlength=C))[::-1][:4]:
    m = (cc == c) & (ss >= 0.001); x = bb[m]
    print("class", c, "n", m.sum(), "width median %.4f height median %.4f  x-range %.3f..%.3f" % (np.median(x[:,2]-x[:,0]), np.median(x[:,3]-x[:,1]), x[:,0].min(), x[:,2].max()))
