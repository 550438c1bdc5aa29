"""Diagnostic (not a test): is the fp16 step's loss consistent with its own heads?  The float64 loss of the oracle evaluated ON the heads
yn_train_forward returns, next to the losses yn_train_step reports and the oracles' (exact, fp16-storage emulation).
python tools/diag_h16_loss.py [S C B backbone]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from yolo_nano_amd import arch, weights, capi
from oracle.torch_port import TrainNet
from tests.test_gpu_train import _targets

S, C, B, bk = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]) if len(sys.argv) > 4 else (128, 20, 8, "1.0x")
sd = weights.make_state_dict(bk, C)
for hd in (1, 2, 3):
    sd["head_det_%d.4.bias" % hd][:3] = -4.6
x = weights.make_input(B, S, seed=21)
t = _targets(S, C, B)
net = TrainNet(sd, bk, C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64)
emu = TrainNet(sd, bk, C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64, fp16_storage=True)
with torch.no_grad():
    h64 = net.forward_raw(x); hq = emu.forward_raw(x)
    print("exact  ", [float(v) for v in net.losses(h64, torch.as_tensor(t).double(), S)])
    print("emul   ", [float(v) for v in net.losses(hq, torch.as_tensor(t).double(), S)])
h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, bk, max_batch=B)
h.load_state_dict(sd); h.train_bind(); h.train_precision("f16")
for rep in range(2):
    heads = [v.permute(0, 3, 1, 2).cpu().double() for v in h.train_forward(torch.as_tensor(x).cuda())]
    with torch.no_grad():
        print("on HIP heads (float64 loss)", [float(v) for v in net.losses(heads, torch.as_tensor(t).double(), S)])
    print("HIP step                   ", h.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(t).cuda(), lr=1e-3, update=False).cpu().numpy().tolist())
rms = lambda a: float(np.sqrt((np.asarray(a, np.float64) ** 2).mean()))
for k in range(3):
    print("head", k, "rms hip-exact", rms(heads[k].numpy() - h64[k].numpy()), "emul-exact", rms(hq[k].numpy() - h64[k].numpy()), "hip-emul", rms(heads[k].numpy() - hq[k].numpy()))
