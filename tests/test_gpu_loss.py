"""GPU parity of the training loss: yn_loss / yn_loss_heads against the reference's own
values + autograd gradients (tests/golden/loss.npz) and against the CPU oracle on seeded inputs.
Tolerance: 1e-4 relative on the four loss values, 1e-4 rel / 1e-6 abs on gradients (fp32)."""
import numpy as np
import pytest
import torch

from oracle import loss as oloss
from yolo_nano_amd import arch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def test_loss_matches_reference_autograd(golden):
    from yolo_nano_amd import capi
    g = golden("loss.npz")
    S, C, B = int(g["S"]), int(g["C"]), int(g["B"])
    h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=B)
    losses, (gc, gcl, gt) = h.loss(dev(g["pred_conf"][..., 0]), dev(g["pred_cls"]), dev(g["pred_txtytwth"]), dev(g["target"]))
    np.testing.assert_allclose(losses.cpu().numpy(), g["losses"], rtol=1e-4)
    np.testing.assert_allclose(gc.cpu().numpy(), g["g_conf"][..., 0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gcl.cpu().numpy(), g["g_cls"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gt.cpu().numpy(), g["g_txtytwth"], rtol=1e-4, atol=1e-6)
    # forward-only call gives the same values
    l2, none = h.loss(dev(g["pred_conf"][..., 0]), dev(g["pred_cls"]), dev(g["pred_txtytwth"]), dev(g["target"]), grads=False)
    assert none is None and torch.equal(l2, losses)
    h.close()


def _random_case(S, C, B, seed, n_pos):
    rs = np.random.RandomState(seed)
    N = arch.num_predictions(S)
    conf = rs.standard_normal((B, N)).astype(np.float32)
    cls = rs.standard_normal((B, N, C)).astype(np.float32)
    t = (rs.standard_normal((B, N, 4)) * 0.5).astype(np.float32)
    target = np.zeros((B, N, 11), np.float32)
    for b in range(B):
        idx = rs.choice(N, n_pos, replace=False)
        target[b, idx, 0] = 1.0
        target[b, idx, 1] = rs.randint(0, C, n_pos)
        target[b, idx, 2:4] = rs.uniform(0, 1, (n_pos, 2))
        target[b, idx, 4:6] = rs.standard_normal((n_pos, 2)) * 0.3
        target[b, idx, 6] = rs.uniform(1.0, 2.0, n_pos)
        c = rs.uniform(0.2, 0.8, (n_pos, 2)); wh = rs.uniform(0.05, 0.4, (n_pos, 2))
        target[b, idx, 7:9], target[b, idx, 9:11] = c - wh / 2, c + wh / 2
        ign = rs.choice(N, n_pos, replace=False)             # ignored anchors: obj = -1, weight = -1 (tools.py:210-211)
        ign = ign[target[b, ign, 0] == 0]
        target[b, ign, 0] = -1.0
        target[b, ign, 6] = -1.0
    return conf, cls, t, target


def test_loss_config3_size_vs_oracle():
    """BASELINE config 3 shape (608x608, COCO head) at a small batch: values + gradients vs the numpy oracle."""
    from yolo_nano_amd import capi
    S, C, B = 608, 80, 2
    conf, cls, t, target = _random_case(S, C, B, 5, 12)
    h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE_COCO, "1.0x", max_batch=B)
    losses, (gc, gcl, gt) = h.loss(dev(conf), dev(cls), dev(t), dev(target))
    ref_l, ref_iou, rc, rcl, rt = oloss.loss_and_grads(conf, cls, t, target, S, arch.MULTI_ANCHOR_SIZE_COCO)
    np.testing.assert_allclose(losses.cpu().numpy(), ref_l, rtol=1e-4)
    np.testing.assert_allclose(gc.cpu().numpy(), rc, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gcl.cpu().numpy(), rcl, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gt.cpu().numpy(), rt, rtol=2e-4, atol=1e-6)
    h.close()


def test_loss_heads_layout_equals_split_layout():
    """yn_loss_heads reads the raw NHWC heads (models/yolo_nano.py:308-330 folded into the kernel): same numbers."""
    from yolo_nano_amd import capi
    S, C, B, A = 128, 20, 3, 3
    conf, cls, t, target = _random_case(S, C, B, 9, 6)
    h = capi.Handle(S, C, arch.MULTI_ANCHOR_SIZE, "1.0x", max_batch=B)
    HC = A * (5 + C)
    heads, off = [], 0
    for s in arch.STRIDES:
        hw = (S // s) ** 2
        hd = np.zeros((B, hw, HC), np.float32)
        sl = slice(off, off + hw * A)
        hd[:, :, :A] = conf[:, sl].reshape(B, hw, A)
        hd[:, :, A:A + A * C] = cls[:, sl].reshape(B, hw, A * C)
        hd[:, :, A + A * C:] = t[:, sl].reshape(B, hw, A * 4)
        heads.append(dev(hd.reshape(B, S // s, S // s, HC)))
        off += hw * A
    l1, (gc, gcl, gt) = h.loss(dev(conf), dev(cls), dev(t), dev(target))
    l2, gh = h.loss_heads(heads, dev(target))
    assert torch.equal(l1, l2)
    off = 0
    for s, g in zip(arch.STRIDES, gh):
        hw = (S // s) ** 2
        g = g.reshape(B, hw, HC)
        sl = slice(off, off + hw * A)
        assert torch.equal(g[:, :, :A].reshape(B, -1), gc[:, sl])
        assert torch.equal(g[:, :, A:A + A * C].reshape(B, hw * A, C), gcl[:, sl])
        assert torch.equal(g[:, :, A + A * C:].reshape(B, hw * A, 4), gt[:, sl])
        off += hw * A
    h.close()


def test_sgd_step_matches_torch_optim():
    """Fused SGD (train.py:167-171: lr, momentum 0.9, weight_decay 5e-4) on a flat bucket of the real parameter count,
    with the 1/world gradient scaling folded in, against torch.optim.SGD on the CPU for three steps."""
    from yolo_nano_amd import capi
    n = arch.param_count("1.0x", 80)                      # 1,326,305 (odd: exercises the scalar tail)
    rs = np.random.RandomState(0)
    p0 = rs.standard_normal(n).astype(np.float32)
    h = capi.Handle(64, 80, arch.MULTI_ANCHOR_SIZE_COCO)
    p = dev(p0.copy()); buf = torch.zeros_like(p)
    ref = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD([ref], lr=1e-3, momentum=0.9, weight_decay=5e-4)
    world = 8
    for step in range(3):
        g = rs.standard_normal(n).astype(np.float32)
        h.sgd_step(p, dev(g * world), buf, 1e-3, 0.9, 5e-4, grad_scale=1.0 / world, first_step=(step == 0))
        ref.grad = torch.from_numpy(g.copy())
        opt.step()
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    h.close()
