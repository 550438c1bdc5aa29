"""CPU restatement of the training loss of the reference.  TEST INFRASTRUCTURE ONLY (see oracle.py).

Restates, in numpy float32 (gradients analytically, same formulas autograd derives):
    models/yolo_nano.py:332-358   decode -> iou -> label assembly -> tools.loss
    tools.py:219-233              iou_score (no epsilon, `en` = all(tl < br))
    tools.py:12-34                MSEWithLogitsLoss (5*pos*(sigmoid-iou)^2 + neg*sigmoid^2, pos: obj==1, neg: obj==0)
    tools.py:236-276              loss: objectness, CE class, BCE txty + MSE twth (weighted, masked), SmoothL1 iou; all sum/B
Pinned against tests/golden/loss.npz (generated from the imported reference incl. autograd gradients).
"""
import numpy as np

from oracle import oracle as orc
from yolo_nano_amd import arch

f32 = np.float32


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)


def decode_norm(txtytwth, S, anchors, A=3):
    """[B,N,4] logits -> xyxy / S  [B,N,4] (unclamped), models/yolo_nano.py:336"""
    B, N, _ = txtytwth.shape
    _, xyxy = orc.decode_boxes(txtytwth.reshape(B, N // A, A, 4), S, anchors, A)
    return (xyxy / f32(S)).astype(f32)


def iou_score(a, b):
    """tools.py:219-233 on [..,4] arrays"""
    tl = np.maximum(a[..., :2], b[..., :2])
    br = np.minimum(a[..., 2:], b[..., 2:])
    area_a = np.prod(a[..., 2:] - a[..., :2], -1)
    area_b = np.prod(b[..., 2:] - b[..., :2], -1)
    en = (tl < br).astype(f32).prod(-1)
    area_i = np.prod(br - tl, -1) * en
    with np.errstate(invalid="ignore", divide="ignore"):
        return (area_i / (area_a + area_b - area_i)).astype(f32)


def loss_and_grads(conf, cls, txtytwth, target, S, anchors, A=3):
    """conf [B,N] / cls [B,N,C] / txtytwth [B,N,4] logits, target [B,N,11] -> (losses[4], iou [B,N], grads of the SUM of
    the four losses w.r.t. conf, cls, txtytwth) — what models/yolo_nano.py:332-358 + train.py:222-229 compute."""
    conf, cls, t, target = (np.asarray(v, dtype=f32) for v in (conf, cls, txtytwth, target))
    B, N, C = cls.shape
    obj, gcls, gt_t, wgt, gt_box = target[..., 0], target[..., 1].astype(np.int64), target[..., 2:6], target[..., 6], target[..., 7:11]
    pred_box = decode_norm(t, S, anchors, A)
    iou = iou_score(pred_box, gt_box)
    pos, neg, mask = (obj == 1).astype(f32), (obj == 0).astype(f32), (obj > 0).astype(f32)
    sg = _sigmoid(conf)
    conf_loss = np.sum(5.0 * pos * (sg - iou) ** 2 + neg * sg ** 2, dtype=np.float64) / B
    mx = cls.max(-1, keepdims=True)
    lse = mx[..., 0] + np.log(np.exp((cls - mx).astype(np.float64)).sum(-1))
    ce = lse - np.take_along_axis(cls, gcls[..., None], -1)[..., 0]
    cls_loss = np.sum(ce * mask, dtype=np.float64) / B
    x = t[..., :2].astype(np.float64)
    bce = np.maximum(x, 0) - x * gt_t[..., :2] + np.log1p(np.exp(-np.abs(x)))
    txty_loss = np.sum(bce.sum(-1) * wgt * mask, dtype=np.float64) / B
    twth_loss = np.sum(((t[..., 2:] - gt_t[..., 2:]).astype(np.float64) ** 2).sum(-1) * wgt * mask, dtype=np.float64) / B
    d = (iou - mask).astype(np.float64)
    sl1 = np.where(np.abs(d) < 1.0, 0.5 * d * d, np.abs(d) - 0.5)
    iou_loss = np.sum(sl1) / B
    losses = np.array([conf_loss, cls_loss, txty_loss + twth_loss, iou_loss], dtype=np.float64)

    # ---- gradients of the sum ----
    g_conf = ((5.0 * pos * 2.0 * (sg - iou) + neg * 2.0 * sg) * sg * (1.0 - sg) / B).astype(f32)
    sm = np.exp((cls - mx).astype(np.float64))
    sm /= sm.sum(-1, keepdims=True)
    onehot = np.zeros_like(sm)
    np.put_along_axis(onehot, gcls[..., None], 1.0, -1)
    g_cls = ((sm - onehot) * mask[..., None] / B).astype(f32)
    g_t = np.zeros(t.shape, dtype=np.float64)
    g_t[..., :2] = (_sigmoid(t[..., :2]) - gt_t[..., :2]) * (wgt * mask)[..., None] / B
    g_t[..., 2:] = 2.0 * (t[..., 2:] - gt_t[..., 2:]) * (wgt * mask)[..., None] / B
    # iou loss path: d sl1/d iou * d iou/d box * d box/d t
    g_iou = np.where(np.abs(d) < 1.0, d, np.sign(d)) / B
    a, b = pred_box.astype(np.float64), gt_box.astype(np.float64)
    tl, br = np.maximum(a[..., :2], b[..., :2]), np.minimum(a[..., 2:], b[..., 2:])
    en = (tl < br).all(-1).astype(np.float64)
    wh_i = br - tl
    I = wh_i.prod(-1) * en
    wa, ha = a[..., 2] - a[..., 0], a[..., 3] - a[..., 1]
    Aa, Ab = wa * ha, (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    U = Aa + Ab - I
    # dI/d(tl), dI/d(br); torch.max/min backward: ties split the gradient in half
    dI_dtl = np.stack([-wh_i[..., 1], -wh_i[..., 0]], -1) * en[..., None]
    dI_dbr = np.stack([wh_i[..., 1], wh_i[..., 0]], -1) * en[..., None]
    w_tl = np.where(a[..., :2] > b[..., :2], 1.0, np.where(a[..., :2] == b[..., :2], 0.5, 0.0))
    w_br = np.where(a[..., 2:] < b[..., 2:], 1.0, np.where(a[..., 2:] == b[..., 2:], 0.5, 0.0))
    dI_da = np.concatenate([dI_dtl * w_tl, dI_dbr * w_br], -1)
    dAa_da = np.stack([-ha, -wa, ha, wa], -1)
    with np.errstate(invalid="ignore", divide="ignore"):
        diou_da = (dI_da * U[..., None] - I[..., None] * (dAa_da - dI_da)) / (U * U)[..., None]
    g_box = np.nan_to_num(diou_da) * g_iou[..., None]                     # d loss / d (x1,y1,x2,y2)/S
    # box = decode(t)/S : x1 = cx - w/2, x2 = cx + w/2 with cx = (sig(tx)+gx)*s, w = exp(tw)*aw
    sxy = _sigmoid(t[..., :2]).astype(np.float64)
    _, stride, anc = orc.create_grid(S, anchors, A)
    stride, anc = stride.reshape(1, N, 2).astype(np.float64), anc.reshape(1, N, 2).astype(np.float64)
    dc_dt = sxy * (1 - sxy) * stride / S
    dwh_dt = np.exp(t[..., 2:].astype(np.float64)) * anc / S
    g_t[..., 0] += (g_box[..., 0] + g_box[..., 2]) * dc_dt[..., 0]
    g_t[..., 1] += (g_box[..., 1] + g_box[..., 3]) * dc_dt[..., 1]
    g_t[..., 2] += 0.5 * (g_box[..., 2] - g_box[..., 0]) * dwh_dt[..., 0]
    g_t[..., 3] += 0.5 * (g_box[..., 3] - g_box[..., 1]) * dwh_dt[..., 1]
    return losses, iou, g_conf, g_cls, g_t.astype(f32)
