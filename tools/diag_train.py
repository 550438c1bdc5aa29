"""Diagnostic (not a test): per-parameter gradient error of the HIP training step and of the fp32 torch port, both
against the fp64 oracle.  python tools/diag_train.py [backbone S C B]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_nano_amd import arch, weights, capi
from oracle.torch_port import TrainNet

backbone, S, C, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("1.0x", 128, 20, 2)
g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train.npz"))
if (S, C, B) == (128, 20, 2):
    target = g["target"]
else:
    rs = np.random.RandomState(5)
    N = arch.num_predictions(S)
    target = np.zeros((B, N, 11), np.float32)
    for b in range(B):
        idx = rs.choice(N, 6, replace=False)
        target[b, idx, 0] = 1.0; target[b, idx, 1] = rs.randint(0, C, 6); target[b, idx, 2:4] = rs.uniform(0, 1, (6, 2))
        target[b, idx, 4:6] = rs.standard_normal((6, 2)) * 0.3; target[b, idx, 6] = rs.uniform(1.0, 2.0, 6)
        c = rs.uniform(0.2, 0.8, (6, 2)); wh = rs.uniform(0.05, 0.4, (6, 2))
        target[b, idx, 7:9], target[b, idx, 9:11] = c - wh / 2, c + wh / 2
x = weights.make_input(B, S, seed=10)
sd = weights.make_state_dict(backbone, C)
for hd in (1, 2, 3):
    sd["head_det_%d.4.bias" % hd][:3] = float(g["init_bias_value"])
res = {}
for dt in (torch.float32, torch.float64):
    net = TrainNet(sd, backbone, C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=dt)
    res[dt] = net.train_step(x, target, S)
h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, backbone, max_batch=B)
h.load_state_dict(sd)
h.train_bind()
losses = h.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(target).cuda(), update=False)
print("losses hip", losses.cpu().numpy(), "f32", res[torch.float32][0], "f64", res[torch.float64][0])
for name, g64 in res[torch.float64][1].items():
    g64 = g64.numpy()
    g32 = res[torch.float32][1][name].double().numpy()
    got = h.flat_grads[h.param_slice(name)].cpu().numpy().reshape(g64.shape).astype(np.float64)
    sc = max(np.abs(g64).max(), 1e-30)
    e_h, e_t = np.abs(got - g64).max() / sc, np.abs(g32 - g64).max() / sc
    flag = " <<<" if e_h > 3 * e_t and e_h > 1e-3 else ""
    print("%-46s %-18s max|g| %9.3g  hip %.2e  torch32 %.2e%s" % (name, tuple(g64.shape), sc, e_h, e_t, flag))
