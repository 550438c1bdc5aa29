"""GPU parity of the training label assigner (tools.multi_gt_creator, tools.py:97-216):
yn_make_targets against the reference's own outputs (tests/golden/targets.npz) and against the oracle at BASELINE
configs[2]'s size.  Everything is integer / exactly-rounded float64 arithmetic and must match BIT-EXACTLY, except
tw/th = log(box/anchor): the device's float64 log may differ from libm's in the last bit, which can move the float32
result by one ulp — those two fields get 1 float32 ulp."""
import numpy as np
import pytest
import torch

from oracle import targets as otg
from yolo_nano_amd import arch

pytestmark = pytest.mark.gpu


def _check(got, ref):
    assert got.shape == ref.shape
    for f in (0, 1, 2, 3, 6, 7, 8, 9, 10):
        np.testing.assert_array_equal(got[..., f], ref[..., f], err_msg="field %d" % f)
    for f in (4, 5):
        ulp = np.spacing(np.abs(ref[..., f]).astype(np.float32))
        assert (np.abs(got[..., f] - ref[..., f]) <= ulp).all(), "field %d" % f


def test_targets_match_reference_fixture(golden):
    from yolo_nano_amd import capi
    g = golden("targets.npz")
    for ci in range(4):
        S, C, B, coco = (int(v) for v in g["case%d_meta" % ci])
        anchors = arch.MULTI_ANCHOR_SIZE_COCO if coco else arch.MULTI_ANCHOR_SIZE
        labels = otg.labels_from_flat(g["case%d_labels" % ci], B)
        h = capi.Handle(S, C, anchors, "1.0x", max_batch=B)
        got = h.make_targets(labels, anchors).cpu().numpy()
        _check(got, g["case%d_target" % ci])
        h.close()


def test_targets_config3_size_vs_oracle_and_shim():
    """608x608, 32 images x 40 objects (+ an empty image): vs the oracle; the module-level shim with the reference's
    signature gives the same tensor; a reused output buffer is fully overwritten."""
    import yolo_nano_amd
    from yolo_nano_amd import capi
    S, C, B = 608, 80, 32
    rs = np.random.RandomState(9)
    labels = []
    for b in range(B):
        n = 0 if b == 5 else 40
        cxy = rs.uniform(0.05, 0.95, (n, 2))
        wh = np.exp(rs.uniform(np.log(0.004), np.log(0.9), (n, 2)))
        box = np.clip(np.concatenate([cxy - wh / 2, cxy + wh / 2], 1), 0, 1).astype(np.float32).astype(np.float64)
        labels.append(np.concatenate([box, rs.randint(0, C, (n, 1)).astype(np.float64)], 1).tolist())
    ref = otg.multi_gt_creator(S, list(arch.STRIDES), labels, arch.MULTI_ANCHOR_SIZE_COCO)
    h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", max_batch=B)
    out = torch.full((B, h.N, 11), 7.0, dtype=torch.float32, device="cuda")
    got = h.make_targets(labels, arch.MULTI_ANCHOR_SIZE_COCO, out=out)
    assert got.data_ptr() == out.data_ptr()
    _check(got.cpu().numpy(), ref)
    t2 = yolo_nano_amd.multi_gt_creator(S, [8, 16, 32], labels, arch.MULTI_ANCHOR_SIZE_COCO)
    assert t2.is_cuda and torch.equal(t2, got)
    # the assigned targets drive a training step end to end
    losses = None
    from yolo_nano_amd import weights
    h.load_state_dict(weights.make_state_dict("1.0x", C))
    h.train_bind()
    x = torch.as_tensor(weights.make_input(2, S, seed=3)).cuda()
    losses = h.train_step(x, got[:2].contiguous(), lr=1e-4)
    assert torch.isfinite(losses).all()
    h.close()
