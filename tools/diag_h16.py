"""Diagnostic (not a test): per-parameter relative L2 error of the HIP fp16 / fp32 training step, of the fp32 oracle and of the
fp16-storage emulation against the fp64 oracle.  python tools/diag_h16.py [S C B backbone]"""
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
rel = lambda a, e: float(np.linalg.norm((a - e).ravel()) / max(np.linalg.norm(e.ravel()), 1e-30))
l64, g64 = TrainNet(sd, bk, C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64).train_step(x, t, S)
g64 = {k: v.numpy() for k, v in g64.items()}
l32, g32 = TrainNet(sd, bk, C, anchors=arch.MULTI_ANCHOR_SIZE).train_step(x, t, S)
lq, gq = TrainNet(sd, bk, C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64, fp16_storage=True).train_step(x, t, S)
res = {}
for dt in ("f32", "f16"):
    h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, bk, max_batch=B)
    h.load_state_dict(sd); h.train_bind(); h.train_precision(dt)
    ls = h.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(t).cuda(), lr=1e-3, update=False)
    res[dt] = (ls.cpu().numpy(), {k: h.flat_grads[h.param_slice(k)].cpu().numpy().reshape(v.shape) for k, v in g64.items()})
    h.close()
print("losses fp64", l64, "\n fp32 oracle", l32, "\n q16 emul", lq, "\n hip f32", res["f32"][0], "\n hip f16", res["f16"][0])
gmax = max(float(np.abs(v).max()) for v in g64.values())
rows = []
for k, e in g64.items():
    if float(np.abs(e).max()) < 1e-9 * gmax:
        continue
    rows.append((k, rel(g32[k].double().numpy(), e), rel(gq[k].numpy(), e), rel(res["f32"][1][k].astype(np.float64), e), rel(res["f16"][1][k].astype(np.float64), e)))
print("%-44s %10s %10s %10s %10s" % ("parameter", "fp32 orc", "q16 emul", "hip f32", "hip f16"))
for r in sorted(rows, key=lambda r: -r[4])[:40]:
    print("%-44s %10.2e %10.2e %10.2e %10.2e" % r)
arr = np.array([r[1:] for r in rows])
print("median", np.median(arr, 0), "\nmax   ", arr.max(0), "\nfinite f16:", bool(np.isfinite(np.concatenate([v.ravel() for v in res['f16'][1].values()])).all()))
