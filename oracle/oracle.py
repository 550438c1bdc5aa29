"""CPU oracle for the YOLO-Nano hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (yolo-nano_amd/) never does.  Arithmetic lives in
``yn_oracle.c`` (plain C, one IEEE binary32 op per source op); this file is the
network wiring, restating the reference's forward graph with numpy arrays:

    backbone/shufflenetv2.py:69-78,157-167   ShuffleV2Block / ShuffleNetV2.forward
    utils/modules.py:8-18                    Conv = conv + BN + LeakyReLU(0.1)
    models/yolo_nano.py:282-301              lateral / FPN / PAN / heads
    models/yolo_nano.py:362-373              score head + postprocess

Pinned against tests/golden/*.npz (generated from the imported reference by
tests/golden/gen_golden.py) in tests/test_oracle_golden.py.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from yolo_nano_amd import arch  # noqa: E402  (pure-python network spec, no torch, no HIP)

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libyn_oracle.so")
        if not os.path.exists(path):
            from oracle import build as _b
            _b.build()
        L = ctypes.CDLL(path)
        L.yo_conv2d.argtypes = [_f32p] + [ctypes.c_int] * 4 + [_f32p, _f32p] + [ctypes.c_int] * 5 + [_f32p]
        L.yo_bn_eval.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _f32p, ctypes.c_float]
        L.yo_act.argtypes = [_f32p, ctypes.c_size_t, ctypes.c_int]
        L.yo_fold_conv_bn.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p, _f32p, ctypes.c_float, _f32p, _f32p]
        L.yo_maxpool3x3s2.argtypes = [_f32p] + [ctypes.c_int] * 4 + [_f32p]
        L.yo_channel_shuffle.argtypes = [_f32p] + [ctypes.c_int] * 4 + [_f32p]
        L.yo_add_up2.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
        L.yo_add_down2.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p]
        L.yo_score_decode.argtypes = [ctypes.POINTER(_f32p), ctypes.c_int, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p]
        L.yo_decode_boxes.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _f32p, _f32p, _f32p]
        L.yo_nms.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_float, ctypes.c_int, _i64p]
        L.yo_nms.restype = ctypes.c_int
        L.yo_postprocess.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                     _f32p, _f32p, _i64p, _i64p]
        L.yo_postprocess.restype = ctypes.c_int
        L.yo_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- primitives -------------------------------------------------------------------------------
def conv2d(x, w, b=None, stride=1, pad=0, groups=1):
    x, w = _c(x), _c(w)
    b = _c(b) if b is not None else None
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = np.empty((B, Cout, Ho, Wo), dtype=np.float32)
    lib().yo_conv2d(_p(x), B, Cin, H, W, _p(w), _p(b), Cout, k, stride, pad, groups, _p(y))
    return y


def bn_eval(x, gamma, beta, mean, var, eps=arch.BN_EPS):
    x = _c(x).copy()
    B, C, H, W = x.shape
    lib().yo_bn_eval(_p(x), B, C, H * W, _p(_c(gamma)), _p(_c(beta)), _p(_c(mean)), _p(_c(var)), eps)
    return x


def act(x, kind):
    x = _c(x).copy()
    lib().yo_act(_p(x), x.size, kind)
    return x


def fold_conv_bn(w, b, gamma, beta, mean, var, eps=arch.BN_EPS):
    """utils/fuse_conv_bn.py:6-22"""
    w = _c(w)
    Cout = w.shape[0]
    wo, bo = np.empty_like(w), np.empty((Cout,), dtype=np.float32)
    lib().yo_fold_conv_bn(_p(w), _p(_c(b)) if b is not None else None, Cout, w.size // Cout,
                          _p(_c(gamma)), _p(_c(beta)), _p(_c(mean)), _p(_c(var)), eps, _p(wo), _p(bo))
    return wo, bo


def maxpool3x3s2(x):
    x = _c(x)
    B, C, H, W = x.shape
    y = np.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=np.float32)
    lib().yo_maxpool3x3s2(_p(x), B, C, H, W, _p(y))
    return y


def channel_shuffle(x, groups=2):
    x = _c(x)
    B, C, H, W = x.shape
    y = np.empty_like(x)
    lib().yo_channel_shuffle(_p(x), B, C, H * W, groups, _p(y))
    return y


def add_up2(a, b):
    a, b = _c(a), _c(b)
    B, C, H, W = a.shape
    y = np.empty_like(a)
    lib().yo_add_up2(_p(a), _p(b), B * C, H, W, _p(y))
    return y


def add_down2(a, b):
    a, b = _c(a), _c(b)
    B, C, H, W = a.shape
    y = np.empty_like(a)
    lib().yo_add_down2(_p(a), _p(b), B * C, H, W, _p(y))
    return y


# ---- network ----------------------------------------------------------------------------------
class Net:
    """The reference's eval-mode network on numpy state dicts (key set = arch.state_dict_spec)."""

    def __init__(self, state_dict, backbone="1.0x", num_classes=20, num_anchors=3, fold=False):
        self.backbone, self.C, self.A = backbone, num_classes, num_anchors
        self.sd = {k: np.asarray(v) for k, v in state_dict.items()}
        self.specs = {s.name: s for s in arch.conv_specs(backbone, num_classes, num_anchors)}
        self.fold = fold
        self._folded = {}

    def folded(self, name):
        """(W', b') of utils/fuse_conv_bn.py for conv `name`."""
        if name not in self._folded:
            sp, sd = self.specs[name], self.sd
            w = sd[sp.conv + ".weight"]
            b = sd.get(sp.conv + ".bias")
            if sp.bn is None:
                self._folded[name] = (_c(w), _c(b))
            else:
                self._folded[name] = fold_conv_bn(w, b, sd[sp.bn + ".weight"], sd[sp.bn + ".bias"],
                                                  sd[sp.bn + ".running_mean"], sd[sp.bn + ".running_var"])
        return self._folded[name]

    def conv(self, name, x):
        sp, sd = self.specs[name], self.sd
        pad = 1 if sp.kind != "pw" else 0
        groups = sp.cout if sp.kind == "dw3" else 1
        if self.fold or sp.bn is None:
            w, b = self.folded(name)
            y = conv2d(x, w, b, sp.stride, pad, groups)
        else:
            y = conv2d(x, sd[sp.conv + ".weight"], sd.get(sp.conv + ".bias"), sp.stride, pad, groups)
            y = bn_eval(y, sd[sp.bn + ".weight"], sd[sp.bn + ".bias"], sd[sp.bn + ".running_mean"], sd[sp.bn + ".running_var"])
        return act(y, sp.act) if sp.act else y

    def block(self, prefix, x, stride):
        """backbone/shufflenetv2.py:69-78"""
        if stride == 1:
            c = x.shape[1] // 2
            x1, x2 = x[:, :c], x[:, c:]
            b2 = self.conv(prefix + ".b2.pw2", self.conv(prefix + ".b2.dw", self.conv(prefix + ".b2.pw1", x2)))
            out = np.concatenate([x1, b2], 1)
        else:
            b1 = self.conv(prefix + ".b1.pw", self.conv(prefix + ".b1.dw", x))
            b2 = self.conv(prefix + ".b2.pw2", self.conv(prefix + ".b2.dw", self.conv(prefix + ".b2.pw1", x)))
            out = np.concatenate([b1, b2], 1)
        return channel_shuffle(out, 2)

    def backbone_forward(self, x):
        """backbone/shufflenetv2.py:157-167 -> (c3, c4, c5)"""
        x = maxpool3x3s2(self.conv("stem", x))
        outs = []
        for si, rep in enumerate(arch.STAGE_REPEATS):
            for bi in range(rep):
                x = self.block("backbone.stage%d.%d" % (si + 2, bi), x, 2 if bi == 0 else 1)
            outs.append(x)
        return tuple(outs)

    def forward_raw(self, x):
        """models/yolo_nano.py:284-301 -> three raw NCHW head tensors [B, A(1+C+4), Hs, Ws]"""
        c3, c4, c5 = self.backbone_forward(_c(x))
        p3, p4, p5 = self.conv("conv1x1_0", c3), self.conv("conv1x1_1", c4), self.conv("conv1x1_2", c5)
        p4 = self.conv("smooth_0", add_up2(p4, p5))
        p3 = self.conv("smooth_1", add_up2(p3, p4))
        p4 = self.conv("smooth_2", add_down2(p4, p3))
        p5 = self.conv("smooth_3", add_down2(p5, p4))
        outs = []
        for h, p in ((1, p3), (2, p4), (3, p5)):
            n = "head_det_%d" % h
            for j in range(5):
                p = self.conv("%s.%d" % (n, j), p)
            outs.append(p)
        return outs


def score_decode(heads_one_image, S, C, anchors, A=3):
    """models/yolo_nano.py:308-330,365-367 for ONE image -> (all_bbox [N,4], all_class [N,C])"""
    hs = [_c(h) for h in heads_one_image]
    N = arch.num_predictions(S, A)
    bbox, cls = np.empty((N, 4), np.float32), np.empty((N, C), np.float32)
    arr = (_f32p * 3)(*[_p(h) for h in hs])
    anc = _c(np.asarray(anchors, dtype=np.float32).reshape(3, A, 2))
    lib().yo_score_decode(arr, S, C, A, _p(anc), _p(bbox), _p(cls))
    return bbox, cls


def decode_boxes(txtytwth, S, anchors, A=3):
    """models/yolo_nano.py:120-156 : [B, HW, A, 4] -> (xywh [B,N,4], xyxy [B,N,4]) in pixels"""
    t = _c(txtytwth)
    B = t.shape[0]
    N = t.shape[1] * t.shape[2]
    anc = _c(np.asarray(anchors, dtype=np.float32).reshape(3, A, 2))
    xywh, xyxy = np.empty((B, N, 4), np.float32), np.empty((B, N, 4), np.float32)
    for b in range(B):
        lib().yo_decode_boxes(_p(t[b]), S, A, _p(anc), _p(xywh[b]), _p(xyxy[b]))
    return xywh, xyxy


def create_grid(S, anchors, A=3):
    """models/yolo_nano.py:86-112 -> (grid [1,HW,1,2], stride [1,HW,A,2], anchors [1,HW,A,2])"""
    anc = np.asarray(anchors, dtype=np.float32).reshape(3, A, 2)
    g, st, aw = [], [], []
    for i, s in enumerate(arch.STRIDES):
        ws = hs = S // s
        gy, gx = np.meshgrid(np.arange(hs), np.arange(ws), indexing="ij")
        g.append(np.stack([gx, gy], -1).astype(np.float32).reshape(1, hs * ws, 1, 2))
        st.append(np.full((1, hs * ws, A, 2), float(s), dtype=np.float32))
        aw.append(np.tile(anc[i][None], (hs * ws, 1, 1)))
    return np.concatenate(g, 1), np.concatenate(st, 1), np.concatenate(aw, 0)[None]


def nms(dets, scores, thresh=0.5, diou=False):
    """models/yolo_nano.py:159-188 / :191-242 -> list of kept indices in pick order"""
    d, s = _c(dets), _c(scores)
    n = len(s)
    keep = np.empty((max(n, 1),), dtype=np.int64)
    k = lib().yo_nms(_p(d), _p(s), n, np.float32(thresh), int(diou), keep.ctypes.data_as(_i64p))
    return keep[:k].tolist()


def postprocess(all_local, all_conf, conf_thresh=0.001, nms_thresh=0.5, diou=False, return_index=False):
    """models/yolo_nano.py:245-279 -> (bboxes [K,4] f32, scores [K] f32, cls_inds [K] i64)"""
    b, c = _c(all_local), _c(all_conf)
    N, C = c.shape
    ob, os_, oc, oi = (np.empty((max(N, 1), 4), np.float32), np.empty((max(N, 1),), np.float32),
                       np.empty((max(N, 1),), np.int64), np.empty((max(N, 1),), np.int64))
    K = lib().yo_postprocess(_p(b), _p(c), N, C, np.float32(conf_thresh), np.float32(nms_thresh), int(diou),
                             _p(ob), _p(os_), oc.ctypes.data_as(_i64p), oi.ctypes.data_as(_i64p))
    out = (ob[:K].copy(), os_[:K].copy(), oc[:K].copy())
    return out + (oi[:K].copy(),) if return_index else out


def infer(net, x, S, anchors, conf_thresh=0.001, nms_thresh=0.5, image=0):
    """Full eval-mode YOLONano.forward for one image of the batch (the reference uses image 0)."""
    heads = net.forward_raw(x)
    bbox, cls = score_decode([h[image] for h in heads], S, net.C, anchors, net.A)
    return postprocess(bbox, cls, conf_thresh, nms_thresh)


def tta_merge(per_forward, num_classes, nms_thresh=0.4):
    """utils/misc.py:112-148: per_forward = [(boxes, scores, labels), ...] in the reference's call order (scale 0, scale 0
    flipped, scale 1, ...); odd entries are mirrored back (:126), everything is concatenated and every class goes through
    nms (utils/misc.py:8-37 — the arithmetic of YOLONano.nms).  -> kept (boxes, scores, labels) in concatenation order."""
    bb, sc, lb = [], [], []
    for i, (b, s, l) in enumerate(per_forward):
        b = np.array(b, dtype=np.float32, copy=True)
        if i & 1:
            b[:, 0::2] = 1.0 - b[:, 2::-2]
        bb.append(b); sc.append(np.asarray(s, dtype=np.float32)); lb.append(np.asarray(l, dtype=np.int64))
    bb, sc, lb = np.concatenate(bb), np.concatenate(sc), np.concatenate(lb)
    keep = np.zeros(len(bb), dtype=np.int64)
    for c in range(num_classes):
        inds = np.where(lb == c)[0]
        if len(inds) == 0:
            continue
        keep[inds[nms(bb[inds], sc[inds], nms_thresh)]] = 1
    k = np.where(keep > 0)[0]
    return bb[k], sc[k], lb[k], k
