import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from yolo_nano_amd import arch, capi, weights
S, B = int(sys.argv[1]), int(sys.argv[2]); bb = sys.argv[3] if len(sys.argv) > 3 else "1.0x"
h = capi.Handle(S, 20, arch.MULTI_ANCHOR_SIZE, bb, 0.001, 0.5, max_batch=B)
h.load_state_dict(weights.make_state_dict(bb, 20)); h.fold_bn()
x = torch.as_tensor(weights.make_input(B, S, seed=S + 3 * B)).cuda()
os.environ["YN_CHAIN_PIPE"] = "0"
ref = [t.clone() for t in h.forward_taps(x)]
os.environ["YN_CHAIN_PIPE"] = "2"
got = [t.clone() for t in h.forward_taps(x)]
for i, (a, b) in enumerate(zip(got, ref)):
    d = (a - b).abs()
    nz = (d > 0)
    print("tap", i, tuple(a.shape), "differing", int(nz.sum()), "of", a.numel(), "max", float(d.max()), "max|ref|", float(b.abs().max()))
    if nz.any():
        idx = nz.nonzero()
        print("  first diffs", idx[:6].tolist(), "last", idx[-3:].tolist())
        # by channel parity / position
        ch = idx[:, -1]
        print("  even-channel diffs", int((ch % 2 == 0).sum()), "odd", int((ch % 2 == 1).sum()))
        print("  channels with diffs:", sorted(set(ch.tolist()))[:40], "count", len(set(ch.tolist())))
        rel = (d[nz] / b[nz].abs().clamp_min(1e-6))
        print("  rel diff median", float(rel.median()), "max", float(rel.max()))
