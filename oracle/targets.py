"""CPU restatement of the training label assigner — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows tools.multi_gt_creator (tools.py:97-216) with compute_iou (tools.py:36-76) and set_anchors (tools.py:79-94):
float64 arithmetic on Python-float labels, objects of an image handled in list order (a later object overwrites the
slot of an earlier one; an 'ignore' write only touches fields 0 and 6), float32 cast at the very end.
Pinned against tests/golden/targets.npz (the reference's own outputs)."""
import math

import numpy as np

IGNORE_THRESH = 0.5                                            # data/config.py:3
STRIDES = (8, 16, 32)


def shape_iou(anchors_wh, box_w, box_h):
    """tools.py:36-76 for boxes centred on the origin: IoU of every anchor [w,h] with the gt [box_w, box_h]."""
    out = []
    for aw, ah in anchors_wh:
        ax1, ay1, ax2, ay2 = 0.0 - aw / 2, 0.0 - ah / 2, 0.0 + aw / 2, 0.0 + ah / 2
        gx1, gy1, gx2, gy2 = 0.0 - box_w / 2, 0.0 - box_h / 2, 0.0 + box_w / 2, 0.0 + box_h / 2
        i_w = min(gx2, ax2) - max(gx1, ax1)
        i_h = min(gy2, ay2) - max(gy1, ay1)
        s_i = i_h * i_w
        u = box_w * box_h + aw * ah - s_i + 1e-20
        out.append(s_i / u)
    return out


def multi_gt_creator(input_size, strides, label_lists, anchor_size):
    """-> float32 [B, N, 11] = obj, cls, tx, ty, tw, th, weight, xmin, ymin, xmax, ymax   (tools.py:97-216)"""
    B = len(label_lists)
    h = w = input_size
    A = len(anchor_size) // len(strides)
    anchors = [(float(a[0]), float(a[1])) for a in anchor_size]
    gt = [np.zeros((B, h // s, w // s, A, 11)) for s in strides]
    for b in range(B):
        for lab in label_lists[b]:
            xmin, ymin, xmax, ymax = (float(v) for v in lab[:4])
            cls = int(lab[4])
            c_x = (xmax + xmin) / 2 * w
            c_y = (ymax + ymin) / 2 * h
            box_w = (xmax - xmin) * w
            box_h = (ymax - ymin) * h
            if box_w < 1. or box_h < 1.:
                continue                                           # tools.py:122-124
            iou = shape_iou(anchors, box_w, box_h)
            best = int(np.argmax(iou))
            above = [i for i, v in enumerate(iou) if v > IGNORE_THRESH]
            for index in (above if above else [best]):
                si = index // A
                ab = index - si * A
                s = strides[si]
                c_x_s, c_y_s = c_x / s, c_y / s
                gx, gy = int(c_x_s), int(c_y_s)
                t = gt[si]
                if index == best:
                    if gy < t.shape[1] and gx < t.shape[2]:
                        pw, ph = anchors[index]
                        t[b, gy, gx, ab, 0] = 1.0
                        t[b, gy, gx, ab, 1] = cls
                        t[b, gy, gx, ab, 2:6] = [c_x_s - gx, c_y_s - gy, math.log(box_w / pw), math.log(box_h / ph)]
                        t[b, gy, gx, ab, 6] = 2.0 - (box_w / w) * (box_h / h)
                        t[b, gy, gx, ab, 7:] = [xmin, ymin, xmax, ymax]
                else:
                    t[b, gy, gx, ab, 0] = -1.0                      # tools.py:206-207: ignored in the objectness loss
                    t[b, gy, gx, ab, 6] = -1.0
    return np.concatenate([t.reshape(B, -1, 11) for t in gt], 1).astype(np.float32)


def labels_from_flat(flat, B):
    """[[b, xmin, ymin, xmax, ymax, cls], ...] (fixture layout) -> list of B label lists"""
    out = [[] for _ in range(B)]
    for row in np.asarray(flat, dtype=np.float64).reshape(-1, 6):
        out[int(row[0])].append([float(v) for v in row[1:]])
    return out


def ema_update(ema, model, updates, decay=0.9999):
    """utils/misc.py:76-86 on one array, float32 arithmetic with torch's rounding sequence: v*d, (1-d)*m, sum."""
    d = decay * (1 - math.exp(-updates / 2000.))
    df, omd = np.float32(d), np.float32(1.0 - d)
    return (ema.astype(np.float32) * df + omd * model.astype(np.float32)).astype(np.float32)
