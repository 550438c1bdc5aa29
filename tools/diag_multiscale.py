"""GPU diagnostic: after real updates, gradients of the stepped handle vs a FRESH handle from its snapshot vs the fp32 / fp64 oracles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from yolo_nano_amd import arch, weights, capi
from tests.test_gpu_train import _handle, _targets, _snapshot
from oracle.torch_port import TrainNet

C, B, S = 20, 4, 128
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
g = np.load("tests/golden/train.npz")
rel = lambda a, e: float(np.linalg.norm((a - e).ravel()) / max(np.linalg.norm(e.ravel()), 1e-30))
for trial in range(3):
    h, sd = _handle(S, C, B, float(g["init_bias_value"]))
    h.train_precision(prec)
    for it in range(3):
        x = weights.make_input(B, S, seed=40 + it); t = _targets(S, C, B, seed=7 + it)
        cur = _snapshot(h, sd)
        hB = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=B)
        hB.load_state_dict(cur); hB.train_bind(); hB.train_precision(prec)
        lB = hB.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(t).cuda(), lr=1e-4, update=False).cpu().numpy()
        gB = hB.flat_grads.clone()
        lA = h.train_step(torch.as_tensor(x).cuda(), torch.as_tensor(t).cuda(), lr=1e-4, update=True).cpu().numpy()
        gA = h.flat_grads.clone()
        l64, g64 = TrainNet(cur, "1.0x", C, anchors=arch.MULTI_ANCHOR_SIZE, dtype=torch.float64).train_step(x, t, S, lr=1e-4)
        l32, g32 = TrainNet(cur, "1.0x", C, anchors=arch.MULTI_ANCHOR_SIZE).train_step(x, t, S, lr=1e-4)
        gmax = max(float(v.abs().max()) for v in g64.values())
        groups = {}
        for n, e in g64.items():
            e = e.numpy()
            if float(np.abs(e).max()) < 1e-9 * gmax:
                continue
            a = gA[h.param_slice(n)].cpu().numpy().reshape(e.shape).astype(np.float64)
            key = n.split(".")[0] + ("." + n.split(".")[1] if n.startswith("backbone") else "")
            groups.setdefault(key, []).append((rel(a, e), rel(g32[n].double().numpy(), e)))
        print("trial %d step %d  A-vs-freshB %.2e  loss rel err hip %.1e o32 %.1e" % (trial, it, float((gA - gB).abs().max() / gB.abs().max()),
              float(np.abs(lA - np.array(l64)).max() / np.abs(l64).max()), float(np.abs(np.array(l32) - np.array(l64)).max() / np.abs(l64).max())))
        print("   " + "  ".join("%s hip %.1e o32 %.1e" % (k, np.median([v[0] for v in vs]), np.median([v[1] for v in vs])) for k, vs in sorted(groups.items())))
        hB.close()
    h.close()
