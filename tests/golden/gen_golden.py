#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the *imported* reference.

Runs only in the development container (needs /root/reference); the GPU box
never runs it.  It imports yjh0410/YOLO-Nano unmodified with the three
process-local shims of SURVEY §8(c), loads the build's deterministic weights
(`yolo_nano_amd.weights`) into it and records inputs/outputs as small .npz
files.  Fixtures are data only — no reference source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))      # data/voc.py:11 imports cv2; never called
np.int = int                                                # models/yolo_nano.py:264 (removed alias)
np.bool = bool
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import torch                                                 # noqa: E402
import torch.nn.functional as F                              # noqa: E402

torch.set_num_threads(8)
torch.backends.mkldnn.enabled = False   # plain ATen fp32 kernels: closest to a textbook sum order

from yolo_nano_amd import arch, weights                      # noqa: E402
import tools as ref_tools                                    # noqa: E402
from backbone.shufflenetv2 import ShuffleNetV2, ShuffleV2Block, channel_shuffle  # noqa: E402
from models.yolo_nano import YOLONano                        # noqa: E402
from utils.fuse_conv_bn import fuse_conv_bn                  # noqa: E402
from utils.modules import Conv                               # noqa: E402
from utils import misc as ref_misc                           # noqa: E402
from copy import deepcopy                                    # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


def t2n(t):
    return t.detach().cpu().numpy()


def ref_model(S, C, anchors, conf=0.001, nms=0.5, seed=0, diou=False):
    m = YOLONano(torch.device("cpu"), input_size=S, num_classes=C, trainable=False,
                 conf_thresh=conf, nms_thresh=nms, anchor_size=anchors, diou_nms=diou)
    sd = weights.make_state_dict("1.0x", C, seed=seed)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return m.eval()


def hook_heads(m):
    store = {}
    hs = [getattr(m, "head_det_%d" % i).register_forward_hook(
        lambda mod, inp, out, i=i: store.__setitem__(i, out.detach().clone())) for i in (1, 2, 3)]
    return store, hs


# ----------------------------------------------------------------------------
def gen_keys():
    out = {}
    for C, anchors, tag in ((20, arch.MULTI_ANCHOR_SIZE, "voc"), (80, arch.MULTI_ANCHOR_SIZE_COCO, "coco")):
        m = YOLONano(torch.device("cpu"), input_size=320, num_classes=C, anchor_size=anchors)
        out[tag] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f)
    print("state_dict_keys.json  voc=%d coco=%d keys" % (len(out["voc"]), len(out["coco"])))


def gen_ops():
    rs = np.random.RandomState(7)
    a = {}

    def rnd(*shape):
        return torch.from_numpy(rs.standard_normal(shape).astype(np.float32))
    # depthwise 3x3 s1 / s2, odd sizes, the network's channel counts
    for tag, C, H, W, s in (("dw_s1", 58, 9, 7, 1), ("dw_s2", 24, 11, 8, 2), ("dw_s1b", 96, 5, 5, 1), ("dw_s2b", 116, 6, 6, 2)):
        x, w, b = rnd(2, C, H, W), rnd(C, 1, 3, 3), rnd(C)
        a[tag + "_x"], a[tag + "_w"], a[tag + "_b"] = t2n(x), t2n(w), t2n(b)
        a[tag + "_y"] = t2n(F.conv2d(x, w, b, stride=s, padding=1, groups=C))
    # pointwise
    for tag, ci, co, H, W in (("pw_a", 24, 58, 6, 5), ("pw_b", 116, 116, 4, 7), ("pw_c", 464, 96, 3, 3), ("pw_d", 96, 255, 5, 4), ("pw_e", 232, 232, 2, 3), ("pw_f", 96, 75, 3, 5)):
        x, w, b = rnd(2, ci, H, W), rnd(co, ci, 1, 1) / np.sqrt(ci), rnd(co)
        a[tag + "_x"], a[tag + "_w"], a[tag + "_b"] = t2n(x), t2n(w), t2n(b)
        a[tag + "_y"] = t2n(F.conv2d(x, w, b))
    # dense 3x3
    for tag, ci, co, H, W, s in (("c3_s2", 3, 24, 12, 10, 2), ("c3_s1", 96, 96, 7, 6, 1), ("c3_s2odd", 3, 24, 9, 13, 2)):
        x, w, b = rnd(2, ci, H, W), rnd(co, ci, 3, 3) / np.sqrt(9 * ci), rnd(co)
        a[tag + "_x"], a[tag + "_w"], a[tag + "_b"] = t2n(x), t2n(w), t2n(b)
        a[tag + "_y"] = t2n(F.conv2d(x, w, b, stride=s, padding=1))
    # maxpool 3x3 s2 p1 (backbone/shufflenetv2.py:116)
    for tag, H, W in (("mp_even", 12, 10), ("mp_odd", 9, 13)):
        x = rnd(2, 24, H, W)
        a[tag + "_x"] = t2n(x)
        a[tag + "_y"] = t2n(F.max_pool2d(x, 3, 2, 1))
    # nearest resample (models/yolo_nano.py:291-296)
    x = rnd(2, 96, 4, 6)
    a["up2_x"], a["up2_y"] = t2n(x), t2n(F.interpolate(x, scale_factor=2.0))
    x = rnd(2, 96, 8, 12)
    a["down_x"], a["down_y"] = t2n(x), t2n(F.interpolate(x, scale_factor=0.5))
    # channel shuffle (backbone/shufflenetv2.py:14-28)
    x = rnd(2, 116, 3, 4)
    a["shuf_x"], a["shuf_y"] = t2n(x), t2n(channel_shuffle(x, 2))
    # activations
    x = rnd(4096)
    a["act_x"], a["relu_y"], a["leaky_y"] = t2n(x), t2n(F.relu(x)), t2n(F.leaky_relu(x, 0.1))
    save("ops.npz", **a)


def _load_block(blk, prefix, sd):
    own = blk.state_dict()
    blk.load_state_dict({k: torch.from_numpy(sd[prefix + "." + k].copy()) for k in own})


def gen_blocks():
    sd = weights.make_state_dict("1.0x", 20, seed=0)
    rs = np.random.RandomState(11)
    a = {}
    # stride-2 block: stage2.0 (24 -> 116); stride-1 block: stage2.1 (116 -> 116)
    b2 = ShuffleV2Block(24, 116, 2).eval()
    _load_block(b2, "backbone.stage2.0", sd)
    x = torch.from_numpy(rs.standard_normal((2, 24, 10, 14)).astype(np.float32))
    a["s2_x"], a["s2_y"] = t2n(x), t2n(b2(x))
    b1 = ShuffleV2Block(116, 116, 1).eval()
    _load_block(b1, "backbone.stage2.1", sd)
    x = torch.from_numpy(rs.standard_normal((2, 116, 7, 5)).astype(np.float32))
    a["s1_x"], a["s1_y"] = t2n(x), t2n(b1(x))
    save("blocks.npz", **a)


def gen_backbone():
    a = {}
    for size in ("1.0x", "0.5x"):
        sd = weights.make_state_dict(size, 20, seed=0)
        bb = ShuffleNetV2(size).eval()
        bb.load_state_dict({k[len("backbone."):]: torch.from_numpy(v.copy())
                            for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
        x = torch.from_numpy(weights.make_input(2, 64, seed=3))
        with torch.no_grad():
            c3, c4, c5 = bb(x)
        t = size.replace(".", "")
        a["c3_" + t], a["c4_" + t], a["c5_" + t] = t2n(c3), t2n(c4), t2n(c5)
    save("backbone.npz", **a)


def gen_fold():
    m = ref_model(320, 20, arch.MULTI_ANCHOR_SIZE)
    f = fuse_conv_bn(deepcopy(m))
    fsd = f.state_dict()
    a = {}
    # a few full tensors + an (abs-sum, sum) checksum for every folded conv
    full = ["backbone.conv1.0", "backbone.stage2.0.branch1.0", "backbone.stage3.4.branch2.5",
            "conv1x1_2.convs.0", "smooth_1.convs.0", "head_det_2.0.convs.0", "head_det_3.4"]
    names, sums = [], []
    for sp in arch.conv_specs("1.0x", 20):
        w, b = t2n(fsd[sp.conv + ".weight"]), t2n(fsd[sp.conv + ".bias"])
        names.append(sp.conv)
        sums.append([np.abs(w).astype(np.float64).sum(), w.astype(np.float64).sum(),
                     np.abs(b).astype(np.float64).sum(), b.astype(np.float64).sum()])
        if sp.conv in full:
            a["W:" + sp.conv], a["b:" + sp.conv] = w, b
    a["names"] = np.array(names)
    a["sums"] = np.array(sums, dtype=np.float64)
    # fused vs unfused logits on one input (<= ~2e-7)
    x = torch.from_numpy(weights.make_input(1, 64, seed=5))
    st, hs = hook_heads(m)
    m.set_grid(64)
    with torch.no_grad():
        m(x)
    unf = [t2n(st[i]) for i in (1, 2, 3)]
    for h in hs:
        h.remove()
    st, hs = hook_heads(f)
    f.set_grid(64)
    with torch.no_grad():
        f(x)
    a["fused_vs_unfused_maxabs"] = np.array([np.abs(t2n(st[i]) - unf[i - 1]).max() for i in (1, 2, 3)])
    save("fold.npz", **a)


def _net_case(name, S, C, anchors, B, seed, conf, nms, full_heads=True, sample=4096):
    m = ref_model(S, C, anchors, conf=conf, nms=nms, seed=0)
    x = torch.from_numpy(weights.make_input(B, S, seed=seed))
    st, hs = hook_heads(m)
    with torch.no_grad():
        bboxes, scores, cls_inds = m(x)
    a = {"S": np.int64(S), "C": np.int64(C), "B": np.int64(B), "input_seed": np.int64(seed),
         "conf_thresh": np.float64(conf), "nms_thresh": np.float64(nms),
         "bboxes": bboxes, "scores": scores, "cls_inds": cls_inds.astype(np.int64)}
    rs = np.random.RandomState(99)
    for i in (1, 2, 3):
        h = t2n(st[i])                                  # [B, A(5+C), H, W]
        if full_heads:
            a["head%d" % i] = h
        else:
            flat = h.reshape(-1)
            idx = rs.randint(0, flat.size, size=sample)
            a["head%d_idx" % i] = idx.astype(np.int64)
            a["head%d_val" % i] = flat[idx]
            a["head%d_shape" % i] = np.array(h.shape, dtype=np.int64)
            a["head%d_sum" % i] = np.array([flat.astype(np.float64).sum(), np.abs(flat).astype(np.float64).sum()])
    # score head on image 0 (models/yolo_nano.py:365-367) recomputed from the hooked tensors
    N = arch.num_predictions(S)
    confs, clss, boxes = [], [], []
    for i in (1, 2, 3):
        p = st[i][:1].permute(0, 2, 3, 1).contiguous().view(1, -1, st[i].shape[1])
        confs.append(p[:, :, :3].contiguous().view(1, -1, 1))
        clss.append(p[:, :, 3:3 + 3 * C].contiguous().view(1, -1, C))
        boxes.append(p[:, :, 3 + 3 * C:].contiguous())
    conf_pred, cls_pred = torch.cat(confs, 1), torch.cat(clss, 1)
    txty = torch.cat(boxes, 1).view(1, -1, 3, 4)
    with torch.no_grad():
        all_obj = torch.sigmoid(conf_pred)[0]
        all_bbox = torch.clamp((m.decode_boxes(txty) / m.input_size)[0], 0., 1.)
        all_class = torch.softmax(cls_pred[0], dim=1) * all_obj
    assert all_bbox.shape == (N, 4) and all_class.shape == (N, C)
    if full_heads:
        a["all_bbox"], a["all_class"] = t2n(all_bbox), t2n(all_class)
    # postprocess again from these arrays must reproduce forward's answer
    b2, s2, c2 = m.postprocess(t2n(all_bbox), t2n(all_class))
    assert np.array_equal(b2, bboxes) and np.array_equal(s2, scores) and np.array_equal(c2, cls_inds)
    # tie check: within a class no two surviving candidates share a score (argsort tie order is unpinned)
    sc = t2n(all_class)
    ci = sc.argmax(1)
    best = sc[np.arange(N), ci]
    for c in range(C):
        v = best[(ci == c) & (best >= np.float32(conf))]
        assert len(np.unique(v)) == len(v), "score tie inside class %d of %s" % (c, name)
    for h in hs:
        h.remove()
    print("   %s: N=%d kept=%d" % (name, N, len(scores)))
    save(name, **a)


def gen_net():
    _net_case("net_voc320.npz", 320, 20, arch.MULTI_ANCHOR_SIZE, 1, 1, 0.001, 0.5)
    _net_case("net_coco128_b2.npz", 128, 80, arch.MULTI_ANCHOR_SIZE_COCO, 2, 2, 0.001, 0.5)
    _net_case("net_coco416.npz", 416, 80, arch.MULTI_ANCHOR_SIZE_COCO, 1, 3, 0.001, 0.5, full_heads=False)
    _net_case("net_coco416_t01.npz", 416, 80, arch.MULTI_ANCHOR_SIZE_COCO, 1, 3, 0.1, 0.45, full_heads=False)


def gen_grid_decode():
    a = {}
    for S in (320, 416, 608):
        m = ref_model(S, 80, arch.MULTI_ANCHOR_SIZE_COCO)
        g, s, w = m.create_grid(S)
        a["grid_%d" % S], a["stride_%d" % S], a["anchor_%d" % S] = t2n(g), t2n(s), t2n(w)
    m = ref_model(96, 80, arch.MULTI_ANCHOR_SIZE_COCO)
    rs = np.random.RandomState(21)
    HW = sum((96 // s) ** 2 for s in arch.STRIDES)
    t = torch.from_numpy((rs.standard_normal((2, HW, 3, 4)) * 1.5).astype(np.float32))
    a["dec_S"] = np.int64(96)
    a["dec_in"] = t2n(t)
    a["dec_xywh"] = t2n(m.decode_xywh(t))
    a["dec_boxes"] = t2n(m.decode_boxes(t))
    save("grid_decode.npz", **a)


def _clustered_boxes(rs, n, n_clusters, jitter):
    centers = rs.uniform(0.2, 0.8, (n_clusters, 2))
    sizes = rs.uniform(0.05, 0.3, (n_clusters, 2))
    k = rs.randint(0, n_clusters, n)
    c = centers[k] + rs.standard_normal((n, 2)) * jitter
    s = sizes[k] * np.exp(rs.standard_normal((n, 2)) * jitter * 2)
    b = np.concatenate([c - s / 2, c + s / 2], 1)
    return np.clip(b, 0, 1).astype(np.float32)


def _unique_scores(rs, n, lo=0.0, hi=1.0):
    while True:
        s = rs.uniform(lo, hi, n).astype(np.float32)
        if len(np.unique(s)) == n:
            return s


def gen_nms():
    m = ref_model(64, 20, arch.MULTI_ANCHOR_SIZE, conf=0.001, nms=0.5)
    md = ref_model(64, 20, arch.MULTI_ANCHOR_SIZE, conf=0.001, nms=0.5, diou=True)
    rs = np.random.RandomState(5)
    a = {}
    cases = {}
    # --- single-class nms(dets, scores) -> pick list -------------------------
    cases["random"] = (_clustered_boxes(rs, 300, 300, 0.0), _unique_scores(rs, 300))
    cases["clusters"] = (_clustered_boxes(rs, 500, 6, 0.02), _unique_scores(rs, 500))
    b = _clustered_boxes(rs, 64, 4, 0.02)
    b[5, 2] = b[5, 0]          # zero width
    b[9, 3] = b[9, 1]          # zero height
    b[20] = b[5]               # duplicate of a zero-area box (0/0 -> NaN path)
    b[33, 2] = b[33, 0]
    b[33, 3] = b[33, 1]        # a point
    cases["zero_area"] = (b, _unique_scores(rs, 64))
    # threshold equality: IoU == 0.5 exactly must be KEPT (ovr <= thr)
    b = np.array([[0, 0, 1, 1], [0, 0, 1, 0.5], [0, 0, 0.5, 1], [0.25, 0.25, 0.75, 0.75],
                  [0, 0, 1, 0.5000001], [0.5, 0.5, 1, 1]], dtype=np.float32)
    cases["thr_equal"] = (b, np.array([0.9, 0.8, 0.7, 0.6, 0.5, 0.4], dtype=np.float32))
    cases["single"] = (np.array([[0.1, 0.1, 0.4, 0.5]], dtype=np.float32), np.array([0.3], dtype=np.float32))
    cases["big"] = (_clustered_boxes(rs, 3000, 40, 0.03), _unique_scores(rs, 3000))
    for k, (boxes, scores) in cases.items():
        keep = m.nms(boxes, scores)
        a["nms_%s_boxes" % k], a["nms_%s_scores" % k] = boxes, scores
        a["nms_%s_keep" % k] = np.array(keep, dtype=np.int64)
        keep_d = md.diou_nms(boxes, scores)
        a["diou_%s_keep" % k] = np.array(keep_d, dtype=np.int64)
        keep_m = ref_misc.nms(boxes, scores, 0.4)          # utils/misc.py:8-37 (TTA's thresholded copy)
        a["nms04_%s_keep" % k] = np.array(keep_m, dtype=np.int64)
    a["nms_cases"] = np.array(sorted(cases))
    # --- postprocess(all_local [N,4], all_conf [N,C]) -> triple ------------------
    pcases = {}
    N, C = 1200, 20
    conf = rs.dirichlet(np.ones(C) * 0.3, N).astype(np.float32) * rs.uniform(0, 1, (N, 1)).astype(np.float32)
    pcases["random"] = (_clustered_boxes(rs, N, 30, 0.03), conf)
    conf = np.full((N, C), 1e-6, dtype=np.float32)
    conf[:, 7] = _unique_scores(rs, N, 0.01, 1.0)          # single class dominant: n_c == N
    pcases["one_class"] = (_clustered_boxes(rs, N, 12, 0.03), conf)
    conf = (rs.uniform(0, 0.0009, (50, C))).astype(np.float32)   # nothing passes 0.001 -> empty
    pcases["empty"] = (_clustered_boxes(rs, 50, 5, 0.02), conf)
    conf = rs.uniform(0, 1, (400, C)).astype(np.float32)
    conf[::3, 4] = 2.0                                     # argmax ties across classes never matter; first-max rule
    conf[::3, 11] = 2.0
    conf[::3, 4] += np.linspace(0, 0.5, len(conf[::3])).astype(np.float32)
    conf[::3, 11] = conf[::3, 4]                           # exact tie between class 4 and 11 -> np.argmax takes 4
    pcases["argmax_tie"] = (_clustered_boxes(rs, 400, 10, 0.03), conf)
    for k, (boxes, conf) in pcases.items():
        bb, ss, cc = m.postprocess(boxes, conf)
        a["pp_%s_boxes" % k], a["pp_%s_conf" % k] = boxes, conf
        a["pp_%s_out_boxes" % k], a["pp_%s_out_scores" % k] = bb, ss
        a["pp_%s_out_cls" % k] = np.asarray(cc, dtype=np.int64)
        bb, ss, cc = md.postprocess(boxes, conf)           # the diou_nms=True model: postprocess with nms_processor = diou_nms (models/yolo_nano.py:21,265-272)
        a["ppd_%s_out_boxes" % k], a["ppd_%s_out_scores" % k] = bb, ss
        a["ppd_%s_out_cls" % k] = np.asarray(cc, dtype=np.int64)
    a["pp_cases"] = np.array(sorted(pcases))
    a["conf_thresh"], a["nms_thresh"] = np.float64(0.001), np.float64(0.5)
    save("nms.npz", **a)


def gen_loss():
    """tools.multi_gt_creator / iou_score / loss (+grads) and one train-mode forward."""
    S, C, B = 128, 20, 2
    anchors = arch.MULTI_ANCHOR_SIZE
    rs = np.random.RandomState(31)
    labels = []
    for b in range(B):
        n = 5 + b
        cxy = rs.uniform(0.2, 0.8, (n, 2))
        wh = rs.uniform(0.05, 0.5, (n, 2))
        box = np.clip(np.concatenate([cxy - wh / 2, cxy + wh / 2], 1), 0, 1)
        cls = rs.randint(0, C, (n, 1)).astype(np.float64)
        labels.append(np.concatenate([box, cls], 1).tolist())
    labels[0].append([0.5, 0.5, 0.503, 0.9, 3.0])                  # 'dirty' box (w < 1 px) is skipped
    tgt = ref_tools.multi_gt_creator(S, list(arch.STRIDES), labels, anchors)
    a = {"S": np.int64(S), "C": np.int64(C), "B": np.int64(B), "target": t2n(tgt)}
    a["labels_flat"] = np.array([[b] + l for b, ls in enumerate(labels) for l in ls], dtype=np.float64)
    N = tgt.shape[1]
    conf = torch.from_numpy(rs.standard_normal((B, N, 1)).astype(np.float32)).requires_grad_()
    cls = torch.from_numpy(rs.standard_normal((B, N, C)).astype(np.float32)).requires_grad_()
    txty = torch.from_numpy((rs.standard_normal((B, N, 4)) * 0.5).astype(np.float32)).requires_grad_()
    m = ref_model(S, C, anchors)
    HW = N // 3
    x1y1x2y2_pred = (m.decode_boxes(txty.view(B, HW, 3, 4)) / S).view(-1, 4)
    gt = tgt[:, :, 7:].view(-1, 4)
    iou = ref_tools.iou_score(x1y1x2y2_pred, gt, batch_size=B)
    with torch.no_grad():
        gt_conf = iou.clone()
    label = torch.cat([gt_conf, tgt[:, :, :7]], dim=2)
    losses = ref_tools.loss(pred_conf=conf, pred_cls=cls, pred_txtytwth=txty, pred_iou=iou, label=label)
    total = sum(losses)
    total.backward()
    a["pred_conf"], a["pred_cls"], a["pred_txtytwth"] = t2n(conf), t2n(cls), t2n(txty)
    a["iou"] = t2n(iou)
    a["losses"] = np.array([float(l) for l in losses], dtype=np.float64)
    a["g_conf"], a["g_cls"], a["g_txtytwth"] = t2n(conf.grad), t2n(cls.grad), t2n(txty.grad)
    save("loss.npz", **a)


def gen_05x():
    """0.5x end-to-end raw heads.  YOLONano refuses '0.5x' (models/yolo_nano.py:35-37), so the
    harness composes the reference's own ShuffleNetV2('0.5x') + Conv blocks following
    models/yolo_nano.py:40-70,284-301 (components pinned by import, top-level wiring restated)."""
    C, S, B = 80, 64, 2
    sd = weights.make_state_dict("0.5x", C, seed=0)
    bb = ShuffleNetV2("0.5x").eval()
    bb.load_state_dict({k[len("backbone."):]: torch.from_numpy(v.copy()) for k, v in sd.items() if k.startswith("backbone.")})
    import torch.nn as nn

    def conv(name, c1, c2, k, p=0, g=1):
        c = Conv(c1, c2, k=k, p=p, g=g).eval()
        c.load_state_dict({kk: torch.from_numpy(sd[name + "." + kk].copy()) for kk in c.state_dict()})
        return c
    lat = [conv("conv1x1_%d" % i, c, 96, 1) for i, c in enumerate(arch.STAGE_CH["0.5x"])]
    sm = [conv("smooth_%d" % i, 96, 96, 3, p=1) for i in range(4)]
    heads = []
    for h in (1, 2, 3):
        n = "head_det_%d" % h
        last = nn.Conv2d(96, arch.head_channels(C), 1)
        last.load_state_dict({kk: torch.from_numpy(sd[n + ".4." + kk].copy()) for kk in last.state_dict()})
        heads.append(nn.Sequential(conv(n + ".0", 96, 96, 3, 1, 96), conv(n + ".1", 96, 96, 1),
                                   conv(n + ".2", 96, 96, 3, 1, 96), conv(n + ".3", 96, 96, 1), last).eval())
    x = torch.from_numpy(weights.make_input(B, S, seed=4))
    with torch.no_grad():
        c3, c4, c5 = bb(x)
        p3, p4, p5 = lat[0](c3), lat[1](c4), lat[2](c5)
        p4 = sm[0](p4 + F.interpolate(p5, scale_factor=2.0))
        p3 = sm[1](p3 + F.interpolate(p4, scale_factor=2.0))
        p4 = sm[2](p4 + F.interpolate(p3, scale_factor=0.5))
        p5 = sm[3](p5 + F.interpolate(p4, scale_factor=0.5))
        outs = [heads[0](p3), heads[1](p4), heads[2](p5)]
    save("net_05x_coco64_b2.npz", S=np.int64(S), C=np.int64(C), B=np.int64(B), input_seed=np.int64(4),
         head1=t2n(outs[0]), head2=t2n(outs[1]), head3=t2n(outs[2]))


def gen_train():
    """One full training step of the reference (train.py:212-231): train-mode forward (BatchNorm batch statistics),
    the four losses, total.backward(), SGD(momentum 0.9, wd 5e-4) — two consecutive steps so that the momentum
    buffer is exercised.  Recorded: inputs, labels, losses, a checksum of every gradient + a few full gradients,
    sampled parameters after each step and the BN running statistics."""
    S, C, B = 128, 20, 2
    anchors = arch.MULTI_ANCHOR_SIZE
    m = ref_model(S, C, anchors)
    m.trainable = True                      # SURVEY §8c recipe: construct with trainable=False, then flip (train.py:140-144)
    m.init_bias()
    m.train()
    rs = np.random.RandomState(77)
    labels = []
    for b in range(B):
        n = 4 + b
        cxy = rs.uniform(0.25, 0.75, (n, 2)); wh = rs.uniform(0.08, 0.45, (n, 2))
        box = np.clip(np.concatenate([cxy - wh / 2, cxy + wh / 2], 1), 0, 1)
        labels.append(np.concatenate([box, rs.randint(0, C, (n, 1)).astype(np.float64)], 1).tolist())
    tgt = ref_tools.multi_gt_creator(S, list(arch.STRIDES), labels, anchors)
    opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=5e-4)
    names = [n for n, _ in m.named_parameters()]
    full = ["backbone.conv1.0.weight", "backbone.stage2.0.branch1.0.weight", "backbone.stage3.2.branch2.5.weight", "backbone.stage4.1.branch2.4.bias",
            "conv1x1_1.convs.0.weight", "smooth_1.convs.0.weight", "smooth_2.convs.1.weight", "head_det_1.2.convs.0.weight", "head_det_2.4.weight", "head_det_3.4.bias"]
    a = {"S": np.int64(S), "C": np.int64(C), "B": np.int64(B), "target": t2n(tgt), "lr": np.float64(1e-3),
         "init_bias_value": np.float64(-np.log((1 - 0.01) / 0.01)), "param_names": np.array(names)}
    rsi = np.random.RandomState(5)
    for step in range(2):
        x = torch.from_numpy(weights.make_input(B, S, seed=10 + step))
        losses = m(x, target=tgt)
        total = sum(losses)
        opt.zero_grad()
        total.backward()
        a["losses_%d" % step] = np.array([float(l.detach()) for l in losses], dtype=np.float64)
        gsum = []
        for n, p in m.named_parameters():
            g = t2n(p.grad).astype(np.float64)
            gsum.append([np.abs(g).sum(), g.sum(), np.sqrt((g * g).sum())])
            if n in full:
                a["grad_%d:%s" % (step, n)] = t2n(p.grad).copy()
        a["grad_sums_%d" % step] = np.array(gsum)
        opt.step()
        sd = m.state_dict()
        for n in full:
            a["param_%d:%s" % (step, n)] = t2n(sd[n]).copy()       # copy: the parameter is updated in place by the next step
        for k in ("backbone.conv1.1", "backbone.stage3.0.branch2.4", "smooth_0.convs.1", "head_det_3.1.convs.1"):
            a["rm_%d:%s" % (step, k)] = t2n(sd[k + ".running_mean"]).copy()
            a["rv_%d:%s" % (step, k)] = t2n(sd[k + ".running_var"]).copy()
    save("train.npz", **a)


def gen_targets():
    """tools.multi_gt_creator (tools.py:97-216) on label lists that exercise every branch: plain positives, several
    anchors above the ignore threshold (ignore writes), two objects in one cell/anchor slot (last writer wins, an ignore
    write on top of a positive and the other way round), sub-pixel 'dirty' boxes, image-border boxes, an empty image."""
    rs = np.random.RandomState(77)
    out = {}
    cases = []
    for ci, (S, C, B, anchors, nobj) in enumerate([(320, 20, 3, arch.MULTI_ANCHOR_SIZE, 12), (416, 80, 4, arch.MULTI_ANCHOR_SIZE_COCO, 25),
                                                    (608, 80, 2, arch.MULTI_ANCHOR_SIZE_COCO, 40), (128, 20, 2, arch.MULTI_ANCHOR_SIZE, 6)]):
        labels = []
        for b in range(B):
            n = nobj + b
            cxy = rs.uniform(0.05, 0.95, (n, 2))
            wh = np.exp(rs.uniform(np.log(0.01), np.log(0.9), (n, 2)))         # log-uniform sizes: every anchor gets used
            box = np.clip(np.concatenate([cxy - wh / 2, cxy + wh / 2], 1), 0, 1)
            box = box.astype(np.float32).astype(np.float64)                       # labels are float32 tensors .tolist()-ed (train.py:210)
            cls = rs.randint(0, C, (n, 1)).astype(np.float64)
            ls = np.concatenate([box, cls], 1).tolist()
            # anchor-shaped boxes (IoU 1 with one anchor, > 0.5 with neighbours) stacked on ONE centre: overwrite order matters
            for k in (0, 4, 8, 3):
                aw, ah = anchors[k]
                c = 0.37 + 0.1 * b
                ls.append([max(c - aw / S / 2, 0.0), max(c - ah / S / 2, 0.0), min(c + aw / S / 2, 1.0), min(c + ah / S / 2, 1.0), float(k % C)])
            ls.append([0.5, 0.5, 0.5 + 0.9 / S, 0.9, 3.0])                        # dirty: w < 1 px
            ls.append([0.0, 0.0, 0.2, 0.3, 1.0])                                  # touches the image corner
            ls.append([0.7, 0.6, 1.0, 1.0, 2.0])                                  # touches the far corner
            labels.append(ls)
        if ci == 1:
            labels[2] = []                                                        # an image without objects
        tgt = ref_tools.multi_gt_creator(S, list(arch.STRIDES), labels, [list(a) for a in anchors])
        flat = np.array([[b] + l for b, ls in enumerate(labels) for l in ls], dtype=np.float64).reshape(-1, 6)
        out["case%d_meta" % ci] = np.array([S, C, B, 0 if anchors is arch.MULTI_ANCHOR_SIZE else 1], dtype=np.int64)
        out["case%d_labels" % ci] = flat
        out["case%d_target" % ci] = t2n(tgt)
        nz = np.count_nonzero(t2n(tgt)[..., 0])
        cases.append((S, B, flat.shape[0], nz, int((t2n(tgt)[..., 0] < 0).sum())))
    print("targets cases (S, B, labels, nonzero obj, ignored):", cases)
    save("targets.npz", **out)


def gen_tta():
    """utils/misc.TestTimeAugmentation on the reference model (1.0x, VOC head, 3 scales x flip): the per-forward detections
    it concatenates and the merged result."""
    S, C = 160, 20
    m = ref_model(S, C, arch.MULTI_ANCHOR_SIZE, conf=0.05, nms=0.5).eval()
    x = torch.from_numpy(weights.make_input(1, S, seed=4))
    per = []
    orig_forward = m.forward

    def rec(xx, target=None):
        out = orig_forward(xx)
        per.append(out)
        return out
    m.forward = rec
    tta = ref_misc.TestTimeAugmentation(num_classes=C, nms_thresh=0.4, scale_range=[128, 192, 32])
    with torch.no_grad():
        bb, sc, lb = tta(x, m)
    out = {"S": np.int64(S), "C": np.int64(C), "n_forwards": np.int64(len(per)), "boxes": bb, "scores": sc, "labels": lb.astype(np.int64)}
    for i, (b_, s_, l_) in enumerate(per):
        out["f%d_boxes" % i], out["f%d_scores" % i], out["f%d_labels" % i] = b_, s_, l_.astype(np.int64)
    print("tta forwards:", [len(p[0]) for p in per], "merged:", len(bb))
    save("tta.npz", **out)


def gen_ema():
    """utils/misc.ModelEMA on a small module: three updates with fresh model weights each time -> the EMA state."""
    torch.manual_seed(3)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
    ema = ref_misc.ModelEMA(m, decay=0.9999, updates=0)
    out = {"init:" + k: t2n(v).copy() for k, v in m.state_dict().items()}
    rs = np.random.RandomState(8)
    for step in range(3):
        with torch.no_grad():
            for k, v in m.state_dict().items():
                if v.dtype.is_floating_point:
                    v.copy_(torch.from_numpy(rs.standard_normal(tuple(v.shape)).astype(np.float32)))
                    out["model%d:%s" % (step, k)] = t2n(v).copy()
        ema.update(m)
    for k, v in ema.ema.state_dict().items():
        out["ema:" + k] = t2n(v).copy()
    out["updates"] = np.int64(ema.updates)
    # a later update count: the decay ramps up
    ema.updates = 5000
    ema.update(m)
    for k, v in ema.ema.state_dict().items():
        out["ema_late:" + k] = t2n(v).copy()
    save("ema.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["keys", "ops", "blocks", "backbone", "fold", "net", "grid_decode", "nms", "loss", "05x", "train", "targets", "ema", "tta"]
    for w in which:
        print("==", w)
        globals()["gen_" + w]()
