import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from yolo_nano_amd import arch, capi, weights
B, S, C = 32, int(os.environ.get("S", "416")), 80
h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=B)
h.load_state_dict(weights.make_state_dict("1.0x", C)); h.fold_bn()
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
x = torch.randn((B, 3, S, S), generator=gen, device="cuda")
heads = h.forward_raw(x)
bbox, cls = h.score_full(heads)
sc, ci = cls.max(-1)
for b in range(B):
    ok = sc[b] >= 0.001
    hist = torch.bincount(ci[b][ok], minlength=C).cpu().numpy()
    print(b, np.sort(hist)[::-1][:6].tolist(), "marked", None)
out = h.infer(x)
print("sparse segments:", h.nms_sweep_segments(B, C))
