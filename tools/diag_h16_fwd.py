"""Diagnostic: train-mode forward raw heads, HIP f16 / f32 vs the fp64 oracle and the fp16-storage emulation."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from yolo_nano_amd import arch, weights, capi
from oracle.torch_port import TrainNet

S, C, B, bk = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]) if len(sys.argv) > 4 else (128, 20, 8, "1.0x")
sd = weights.make_state_dict(bk, C)
x = weights.make_input(B, S, seed=21)
with torch.no_grad():
    r64 = [t.numpy() for t in TrainNet(sd, bk, C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64).forward_raw(x)]
    rq = [t.numpy() for t in TrainNet(sd, bk, C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64, fp16_storage=True).forward_raw(x)]
for dt in ("f32", "f16"):
    h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, bk, max_batch=B)
    h.load_state_dict(sd); h.train_bind(); h.train_precision(dt)
    got = [t.permute(0, 3, 1, 2).cpu().numpy() for t in h.train_forward(torch.as_tensor(x).cuda())]
    for k in range(3):
        e64 = np.abs(got[k] - r64[k]); eq = np.abs(got[k] - rq[k])
        print(dt, "head", k, "rms", float(np.sqrt((r64[k] ** 2).mean())), "vs fp64: max %.3e rms %.3e" % (e64.max(), np.sqrt((e64 ** 2).mean())),
              "| vs q16 emul: max %.3e rms %.3e" % (eq.max(), np.sqrt((eq ** 2).mean())), "| emul vs fp64 rms %.3e" % np.sqrt(((rq[k] - r64[k]) ** 2).mean()))
    h.close()
