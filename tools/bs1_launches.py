import sys, torch
sys.path.insert(0, '.')
from yolo_nano_amd import capi, arch, weights
for S in (416, 608):
    h = capi.Handle(S, 80, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", 0.001, 0.5, max_batch=1)
    h.load_state_dict(weights.make_state_dict("1.0x", 80)); h.fold_bn()
    x = torch.as_tensor(weights.make_input(1, S, seed=5)).cuda()
    for _ in range(3): h.infer(x)
    h.profile_enable(True); h.infer(x); recs = h.profile_records(); h.profile_enable(False)
    tot = 0
    for r in recs:
        print(S, "%-40s %-44s %7.1f us" % (r[0][:40], r[1][:44], r[2] * 1e3)); tot += r[2] * 1e3
    print(S, "launches", len(recs), "sum us", round(tot, 1))
