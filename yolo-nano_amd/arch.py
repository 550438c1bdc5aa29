"""Static description of the YOLO-Nano network (torch-free).

This is the single place where the layer list, the channel widths and the
reference's ``state_dict`` key names live.  The HIP library, the oracle, the
deterministic weight generator and the ``YOLONano`` host shim all build from
it, so that the 469 keys of the reference checkpoint format
(``/root/reference/models/yolo_nano.py:29-70`` +
``/root/reference/backbone/shufflenetv2.py:81-129``) are spelled exactly once.

Nothing here imports torch or the reference.
"""

STRIDES = (8, 16, 32)                      # models/yolo_nano.py:23
NECK_CH = 96                               # models/yolo_nano.py:40-47
STAGE_REPEATS = (4, 8, 4)                  # backbone/shufflenetv2.py:90
STEM_CH = 24                               # backbone/shufflenetv2.py:95-102 ([0])
STAGE_CH = {                               # backbone/shufflenetv2.py:95-102
    "0.5x": (48, 96, 192),
    "1.0x": (116, 232, 464),
    "1.5x": (176, 352, 704),
    "2.0x": (244, 488, 976),
}
BN_EPS = 1e-5                              # nn.BatchNorm2d default

# data/config.py:11-17 (pixels, [w, h])
MULTI_ANCHOR_SIZE = [[30.65, 39.12], [50.3, 102.62], [94.98, 64.55],
                     [93.5, 177.51], [165.25, 113.85], [161.83, 240.95],
                     [304.64, 150.34], [251.28, 306.53], [369.38, 261.55]]
MULTI_ANCHOR_SIZE_COCO = [[11.89, 14.24], [30.14, 35.62], [45.99, 87.04],
                          [92.23, 44.43], [130.78, 99.73], [78.99, 170.81],
                          [290.39, 123.89], [165.27, 233.33], [332.57, 279.8]]

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2    # LeakyReLU slope 0.1 (utils/modules.py:14)


class ConvSpec:
    """One convolution of the network, with the BN that follows it (if any).

    ``conv``  state-dict prefix of the nn.Conv2d (``<conv>.weight`` [, ``.bias``])
    ``bn``    state-dict prefix of the BatchNorm2d after it, or None
    ``kind``  'dense3' | 'dw3' | 'pw'
    """
    __slots__ = ("name", "conv", "bn", "kind", "cin", "cout", "stride", "has_bias", "act")

    def __init__(self, name, conv, bn, kind, cin, cout, stride, has_bias, act):
        self.name, self.conv, self.bn, self.kind = name, conv, bn, kind
        self.cin, self.cout, self.stride = cin, cout, stride
        self.has_bias, self.act = has_bias, act

    @property
    def weight_shape(self):
        if self.kind == "dense3":
            return (self.cout, self.cin, 3, 3)
        if self.kind == "dw3":
            return (self.cout, 1, 3, 3)
        return (self.cout, self.cin, 1, 1)

    @property
    def fan_in(self):
        return {"dense3": 9 * self.cin, "dw3": 9, "pw": self.cin}[self.kind]

    def __repr__(self):
        return "ConvSpec(%s %s %d->%d s%d)" % (self.name, self.kind, self.cin, self.cout, self.stride)


def head_channels(num_classes, num_anchors=3):
    return num_anchors * (1 + num_classes + 4)          # models/yolo_nano.py:55


def num_predictions(input_size, num_anchors=3):
    return sum(num_anchors * (input_size // s) ** 2 for s in STRIDES)


def conv_specs(backbone="1.0x", num_classes=20, num_anchors=3):
    """All 77 convolutions in state-dict order."""
    if backbone not in STAGE_CH:
        raise ValueError("unknown backbone width %r" % (backbone,))
    specs = []
    add = specs.append
    # stem: backbone/shufflenetv2.py:109-113 (no conv bias, BN, ReLU)
    add(ConvSpec("stem", "backbone.conv1.0", "backbone.conv1.1", "dense3", 3, STEM_CH, 2, False, ACT_RELU))
    cin = STEM_CH
    for si, (rep, cout) in enumerate(zip(STAGE_REPEATS, STAGE_CH[backbone])):
        st = "backbone.stage%d" % (si + 2)
        bf = cout // 2
        for bi in range(rep):
            p = "%s.%d" % (st, bi)
            stride = 2 if bi == 0 else 1
            if bi == 0:
                # branch1: dw s2 + BN ; pw + BN + ReLU     backbone/shufflenetv2.py:42-49
                add(ConvSpec(p + ".b1.dw", p + ".branch1.0", p + ".branch1.1", "dw3", cin, cin, 2, False, ACT_NONE))
                add(ConvSpec(p + ".b1.pw", p + ".branch1.2", p + ".branch1.3", "pw", cin, bf, 1, False, ACT_RELU))
            b2in = cin if bi == 0 else bf
            # branch2: pw+BN+ReLU ; dw+BN ; pw+BN+ReLU       backbone/shufflenetv2.py:53-63
            add(ConvSpec(p + ".b2.pw1", p + ".branch2.0", p + ".branch2.1", "pw", b2in, bf, 1, False, ACT_RELU))
            add(ConvSpec(p + ".b2.dw", p + ".branch2.3", p + ".branch2.4", "dw3", bf, bf, stride, False, ACT_NONE))
            add(ConvSpec(p + ".b2.pw2", p + ".branch2.5", p + ".branch2.6", "pw", bf, bf, 1, False, ACT_RELU))
        cin = cout
    c3, c4, c5 = STAGE_CH[backbone]
    # neck: utils/modules.py:8-18 Conv = conv(bias=True)+BN+LeakyReLU(0.1); models/yolo_nano.py:40-47
    for i, c in enumerate((c3, c4, c5)):
        n = "conv1x1_%d" % i
        add(ConvSpec(n, n + ".convs.0", n + ".convs.1", "pw", c, NECK_CH, 1, True, ACT_LEAKY))
    for i in range(4):
        n = "smooth_%d" % i
        add(ConvSpec(n, n + ".convs.0", n + ".convs.1", "dense3", NECK_CH, NECK_CH, 1, True, ACT_LEAKY))
    # heads: models/yolo_nano.py:50-70
    hc = head_channels(num_classes, num_anchors)
    for h in (1, 2, 3):
        n = "head_det_%d" % h
        for j, kind in enumerate(("dw3", "pw", "dw3", "pw")):
            add(ConvSpec("%s.%d" % (n, j), "%s.%d.convs.0" % (n, j), "%s.%d.convs.1" % (n, j),
                         kind, NECK_CH, NECK_CH, 1, True, ACT_LEAKY))
        add(ConvSpec(n + ".4", n + ".4", None, "pw", NECK_CH, hc, 1, True, ACT_NONE))
    return specs


def state_dict_spec(backbone="1.0x", num_classes=20, num_anchors=3):
    """[(key, shape, dtype-name)] in the order torch's ``state_dict()`` yields them."""
    out = []
    for s in conv_specs(backbone, num_classes, num_anchors):
        out.append((s.conv + ".weight", s.weight_shape, "float32"))
        if s.has_bias:
            out.append((s.conv + ".bias", (s.cout,), "float32"))
        if s.bn is not None:
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                out.append(("%s.%s" % (s.bn, leaf), (s.cout,), "float32"))
            out.append((s.bn + ".num_batches_tracked", (), "int64"))
    return out


def param_count(backbone="1.0x", num_classes=20, num_anchors=3):
    """Trainable elements (conv weight/bias + BN weight/bias)."""
    n = 0
    for key, shape, _ in state_dict_spec(backbone, num_classes, num_anchors):
        if key.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        k = 1
        for d in shape:
            k *= d
        n += k
    return n


def conv_flops(input_size, backbone="1.0x", num_classes=80):
    """2*MAC over all convolutions, per image."""
    total = 0
    for sp in conv_specs(backbone, num_classes):
        total += 2 * sp.fan_in * sp.cout * _out_pixels(sp, input_size)
    return total


def activation_elements(input_size, backbone="1.0x", num_classes=80):
    """Elements every convolution reads (its input, once) plus writes (its output, once), per image: the layer-wise activation
    traffic behind the HBM floors of DESIGN 4 / 9 - 20.5 M at 416x416, 43.7 M at 608x608 (1.0x, COCO head)."""
    total = 0
    for sp in conv_specs(backbone, num_classes):
        po = _out_pixels(sp, input_size)
        total += sp.cout * po + sp.cin * po * sp.stride * sp.stride
    return total


def _out_pixels(sp, S):
    n = sp.name
    if n == "stem":
        return (S // 2) ** 2
    if n.startswith("backbone.stage"):
        si = int(n[len("backbone.stage")]) - 2          # 0,1,2
        bi = int(n.split(".")[2])
        out_side = S // (8 << si)
        if bi == 0 and n.endswith("b2.pw1"):
            return (out_side * 2) ** 2                  # full-resolution pw of the stride-2 block
        return out_side ** 2
    if n.startswith("conv1x1_"):
        return (S // STRIDES[int(n[-1])]) ** 2
    if n.startswith("smooth_"):
        return (S // STRIDES[(1, 0, 1, 2)[int(n[-1])]]) ** 2
    if n.startswith("head_det_"):
        return (S // STRIDES[int(n[len("head_det_")]) - 1]) ** 2
    raise KeyError(n)
