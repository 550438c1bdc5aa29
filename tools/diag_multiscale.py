"""GPU diagnostic: gradients after set_grid() on a trained handle vs a fresh handle at the new size vs the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolo_nano_amd import arch, weights, capi
from tests.test_gpu_train import _handle, _targets, _snapshot
from oracle.torch_port import TrainNet

C, B = 20, 4
g = np.load("tests/golden/train.npz")
h, sd = _handle(128, C, B, float(g["init_bias_value"]))
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
h.train_precision(prec)
for it in range(2):
    x = weights.make_input(B, 128, seed=40 + it); t = _targets(128, C, B, seed=7 + it)
    h.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(t).cuda(), lr=1e-4, update=True)
h.set_grid(192)
cur = _snapshot(h, sd)
x = weights.make_input(B, 192, seed=42); t = _targets(192, C, B, seed=9)
lA = h.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(t).cuda(), lr=1e-4, update=False).cpu().numpy()
gA = h.flat_grads.clone()
hB = capi.Handle(192, C, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=B)
hB.load_state_dict(cur); hB.train_bind(); hB.train_precision(prec)
lB = hB.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(t).cuda(), lr=1e-4, update=False).cpu().numpy()
gB = hB.flat_grads.clone()
print("losses A", lA, "B", lB)
print("max |gA-gB| / max|gB|", float((gA - gB).abs().max() / gB.abs().max()))
l64, g64 = TrainNet(cur, "1.0x", C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64).train_step(x, t, 192, lr=1e-4)
l32, g32 = TrainNet(cur, "1.0x", C, anchors=arch.MULTI_ANCHOR_SIZE).train_step(x, t, 192, lr=1e-4)
print("l64", l64, "l32", l32)
rel = lambda a, e: float(np.linalg.norm((a - e).ravel()) / max(np.linalg.norm(e.ravel()), 1e-30))
rows = []
gmax = max(float(v.abs().max()) for v in g64.values())
for n, e in g64.items():
    e = e.numpy()
    if float(np.abs(e).max()) < 1e-9 * gmax:
        continue
    a = gA[h.param_slice(n)].cpu().numpy().reshape(e.shape).astype(np.float64)
    b = gB[hB.param_slice(n)].cpu().numpy().reshape(e.shape).astype(np.float64)
    rows.append((rel(a, e), rel(b, e), rel(g32[n].double().numpy(), e), n))
rows.sort(reverse=True)
for r in rows[:25]:
    print("A %.3e  B %.3e  oracle32 %.3e  %s" % r)
print("median A %.3e B %.3e o32 %.3e" % tuple(np.median([r[i] for r in rows]) for i in range(3)))
