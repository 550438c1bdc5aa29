"""PyTorch-CPU restatement of the reference's eval-mode network.  TEST INFRASTRUCTURE ONLY.

Same role and same restrictions as oracle.py (only tests/, smoke() and bench.py's cpu_baseline leg
import it).  It exists because the reference itself runs on torch's CPU kernels: timing THIS port
(all host cores) is the honest CPU baseline for bench.py (`cpu_baseline.kind = "port"`), and it
cross-checks the plain-C oracle.  Wiring follows models/yolo_nano.py:282-301 and
backbone/shufflenetv2.py:69-78,157-167; the BN folding is utils/fuse_conv_bn.py:17-21.
Pinned against tests/golden in tests/test_oracle_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from yolo_nano_amd import arch


class TorchNet:
    def __init__(self, state_dict, backbone="1.0x", num_classes=20, num_anchors=3):
        self.C, self.A, self.backbone = num_classes, num_anchors, backbone
        sd = {k: torch.as_tensor(np.asarray(v)) for k, v in state_dict.items()}
        self.specs = {s.name: s for s in arch.conv_specs(backbone, num_classes, num_anchors)}
        self.wb = {}
        for s in self.specs.values():
            w = sd[s.conv + ".weight"].float()
            b = sd.get(s.conv + ".bias")
            if s.bn is not None and (s.bn + ".weight") in sd:
                f = sd[s.bn + ".weight"] / torch.sqrt(sd[s.bn + ".running_var"] + arch.BN_EPS)
                b0 = b if b is not None else torch.zeros_like(f)
                w = w * f.reshape(-1, 1, 1, 1)
                b = (b0 - sd[s.bn + ".running_mean"]) * f + sd[s.bn + ".bias"]
            self.wb[s.name] = (w.contiguous(), b.contiguous())

    def conv(self, name, x):
        s = self.specs[name]
        w, b = self.wb[name]
        y = F.conv2d(x, w, b, stride=s.stride, padding=0 if s.kind == "pw" else 1, groups=s.cout if s.kind == "dw3" else 1)
        if s.act == arch.ACT_RELU:
            return F.relu_(y)
        if s.act == arch.ACT_LEAKY:
            return F.leaky_relu_(y, 0.1)
        return y

    def block(self, p, x, stride):
        if stride == 1:
            x1, x2 = x.chunk(2, dim=1)
            out = torch.cat((x1, self.conv(p + ".b2.pw2", self.conv(p + ".b2.dw", self.conv(p + ".b2.pw1", x2)))), 1)
        else:
            b1 = self.conv(p + ".b1.pw", self.conv(p + ".b1.dw", x))
            b2 = self.conv(p + ".b2.pw2", self.conv(p + ".b2.dw", self.conv(p + ".b2.pw1", x)))
            out = torch.cat((b1, b2), 1)
        B, C, H, W = out.shape
        return out.view(B, 2, C // 2, H, W).transpose(1, 2).contiguous().view(B, C, H, W)

    @torch.no_grad()
    def forward_raw(self, x):
        x = torch.as_tensor(x).float()
        x = F.max_pool2d(self.conv("stem", x), 3, 2, 1)
        feats = []
        for si, rep in enumerate(arch.STAGE_REPEATS):
            for bi in range(rep):
                x = self.block("backbone.stage%d.%d" % (si + 2, bi), x, 2 if bi == 0 else 1)
            feats.append(x)
        p3, p4, p5 = (self.conv("conv1x1_%d" % i, f) for i, f in enumerate(feats))
        p4 = self.conv("smooth_0", p4 + F.interpolate(p5, scale_factor=2.0))
        p3 = self.conv("smooth_1", p3 + F.interpolate(p4, scale_factor=2.0))
        p4 = self.conv("smooth_2", p4 + F.interpolate(p3, scale_factor=0.5))
        p5 = self.conv("smooth_3", p5 + F.interpolate(p4, scale_factor=0.5))
        outs = []
        for h, p in ((1, p3), (2, p4), (3, p5)):
            for j in range(5):
                p = self.conv("head_det_%d.%d" % (h, j), p)
            outs.append(p)
        return outs

    @torch.no_grad()
    def score_head(self, heads, S, anchors, image=0):
        """models/yolo_nano.py:308-330,362-367 for one image -> numpy (all_bbox [N,4], all_class [N,C])"""
        A, C = self.A, self.C
        anc = torch.tensor(anchors, dtype=torch.float32).view(3, A, 2)
        boxes, clss = [], []
        for si, (h, st) in enumerate(zip(heads, arch.STRIDES)):
            Hs = S // st
            p = h[image].permute(1, 2, 0).reshape(Hs * Hs, -1)
            obj = torch.sigmoid(p[:, :A]).reshape(-1, 1)
            cls = torch.softmax(p[:, A:A + A * C].reshape(-1, C), dim=1) * obj
            t = p[:, A * (1 + C):].reshape(Hs * Hs, A, 4)
            gy, gx = torch.meshgrid(torch.arange(Hs), torch.arange(Hs), indexing="ij")
            g = torch.stack([gx, gy], -1).float().view(-1, 1, 2)
            cxy = (torch.sigmoid(t[..., :2]) + g) * st
            wh = torch.exp(t[..., 2:]) * anc[si][None]
            b = torch.cat([cxy - wh / 2, cxy + wh / 2], -1).view(-1, 4)
            boxes.append(torch.clamp(b / S, 0., 1.))
            clss.append(cls)
        return torch.cat(boxes, 0).numpy(), torch.cat(clss, 0).numpy()


class _Q16(torch.autograd.Function):
    """Storage rounding of the fp16 training step (kernels_h16.hip), emulated: the forward value is rounded to IEEE fp16, and the
    gradient flowing back through the same point is rounded to fp16 while it carries the loss scale (a power of two: the rounding
    is that of the scaled value, the result is returned unscaled).  Identity otherwise."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.to(torch.float16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        s = ctx.scale
        return ((g * s).to(torch.float16).to(g.dtype) / s), None


class _W16(torch.autograd.Function):
    """fp16 copy of an fp32 master weight as the f16 MFMA sees it; the gradient goes straight through to the master weight."""

    @staticmethod
    def forward(ctx, w):
        return w.to(torch.float16).to(w.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class TrainNet:
    """Train-mode restatement (torch autograd on the CPU): BatchNorm with batch statistics (momentum 0.1, eps 1e-5,
    running-stat update), the loss of tools.py:219-276 / models/yolo_nano.py:332-358, SGD(momentum, weight_decay) as
    train.py:167-171,222-231.  Pinned against tests/golden/train.npz (two reference training steps)."""

    def __init__(self, state_dict, backbone="1.0x", num_classes=20, num_anchors=3, anchors=None, dtype=torch.float32, fp16_storage=False, loss_scale=1024.0):
        """dtype=torch.float64 gives the round-off-free gradient the fp32 paths (reference, this port, HIP) are judged against.
        fp16_storage=True emulates the fp16 training step's storage points (conv outputs, BN+activation outputs, their gradients,
        the GEMM weights) in otherwise exact arithmetic (use with dtype=float64): what fp16 STORAGE alone does to the gradients —
        the yardstick the HIP fp16 step is judged by, as the fp32 oracle run is for the fp32 step."""
        self.C, self.A, self.backbone, self.dt = num_classes, num_anchors, backbone, dtype
        self.q16, self.loss_scale = bool(fp16_storage), float(loss_scale)
        self.anchors = torch.tensor(anchors, dtype=dtype).view(3, num_anchors, 2)
        self.specs = [s for s in arch.conv_specs(backbone, num_classes, num_anchors)]
        self.by = {s.name: s for s in self.specs}
        self.p = {}
        for k, v in state_dict.items():
            t = torch.as_tensor(np.asarray(v)).clone()
            if t.dtype == torch.float32:
                t = t.to(dtype)
                if not k.endswith(("running_mean", "running_var")):
                    t.requires_grad_(True)
            self.p[k] = t
        self.momentum_buf = {}

    def params(self):
        return {k: v for k, v in self.p.items() if v.requires_grad}

    def conv(self, name, x):
        s, p = self.by[name], self.p
        w = p[s.conv + ".weight"]
        if self.q16 and s.kind != "dw3" and name != "stem":      # the GEMM-shaped convs read fp16 packs; depthwise / stem taps stay fp32
            w = _W16.apply(w)
        y = F.conv2d(x, w, p.get(s.conv + ".bias"), stride=s.stride,
                     padding=0 if s.kind == "pw" else 1, groups=s.cout if s.kind == "dw3" else 1)
        if self.q16:
            y = _Q16.apply(y, self.loss_scale)
        if s.bn is None:
            return y
        y = F.batch_norm(y, p[s.bn + ".running_mean"], p[s.bn + ".running_var"], p[s.bn + ".weight"], p[s.bn + ".bias"],
                         training=True, momentum=0.1, eps=arch.BN_EPS)
        if s.act == arch.ACT_RELU:
            y = F.relu(y)
        elif s.act == arch.ACT_LEAKY:
            y = F.leaky_relu(y, 0.1)
        return _Q16.apply(y, self.loss_scale) if self.q16 else y

    def block(self, pfx, x, stride):
        if stride == 1:
            x1, x2 = x.chunk(2, dim=1)
            out = torch.cat((x1, self.conv(pfx + ".b2.pw2", self.conv(pfx + ".b2.dw", self.conv(pfx + ".b2.pw1", x2)))), 1)
        else:
            b1 = self.conv(pfx + ".b1.pw", self.conv(pfx + ".b1.dw", x))
            b2 = self.conv(pfx + ".b2.pw2", self.conv(pfx + ".b2.dw", self.conv(pfx + ".b2.pw1", x)))
            out = torch.cat((b1, b2), 1)
        B, C, H, W = out.shape
        return out.view(B, 2, C // 2, H, W).transpose(1, 2).contiguous().view(B, C, H, W)

    def forward_raw(self, x):
        x = F.max_pool2d(self.conv("stem", torch.as_tensor(x).to(self.dt)), 3, 2, 1)
        feats = []
        for si, rep in enumerate(arch.STAGE_REPEATS):
            for bi in range(rep):
                x = self.block("backbone.stage%d.%d" % (si + 2, bi), x, 2 if bi == 0 else 1)
            feats.append(x)
        q = (lambda t: _Q16.apply(t, self.loss_scale)) if self.q16 else (lambda t: t)       # the FPN / PAN sums are stored tensors too
        p3, p4, p5 = (self.conv("conv1x1_%d" % i, f) for i, f in enumerate(feats))
        p4 = self.conv("smooth_0", q(p4 + F.interpolate(p5, scale_factor=2.0)))
        p3 = self.conv("smooth_1", q(p3 + F.interpolate(p4, scale_factor=2.0)))
        p4 = self.conv("smooth_2", q(p4 + F.interpolate(p3, scale_factor=0.5)))
        p5 = self.conv("smooth_3", q(p5 + F.interpolate(p4, scale_factor=0.5)))
        outs = []
        for h, pp in ((1, p3), (2, p4), (3, p5)):
            for j in range(5):
                pp = self.conv("head_det_%d.%d" % (h, j), pp)
            outs.append(pp)
        return outs

    def losses(self, heads, target, S):
        """models/yolo_nano.py:308-358 + tools.py:219-276 in torch (autograd-able)."""
        A, C = self.A, self.C
        B = heads[0].shape[0]
        confs, clss, ts, grids, strides, ancs = [], [], [], [], [], []
        for si, (h, st) in enumerate(zip(heads, arch.STRIDES)):
            Hs = S // st
            p = h.permute(0, 2, 3, 1).reshape(B, Hs * Hs, -1)
            confs.append(p[:, :, :A].reshape(B, -1))
            clss.append(p[:, :, A:A + A * C].reshape(B, -1, C))
            ts.append(p[:, :, A * (1 + C):].reshape(B, -1, 4))
            gy, gx = torch.meshgrid(torch.arange(Hs), torch.arange(Hs), indexing="ij")
            g = torch.stack([gx, gy], -1).to(self.dt).view(-1, 1, 2).repeat(1, A, 1).view(-1, 2)
            grids.append(g); strides.append(torch.full((Hs * Hs * A, 1), float(st), dtype=self.dt)); ancs.append(self.anchors[si].repeat(Hs * Hs, 1))
        conf, cls, t = torch.cat(confs, 1), torch.cat(clss, 1), torch.cat(ts, 1)
        grid, stride, anc = torch.cat(grids, 0), torch.cat(strides, 0), torch.cat(ancs, 0)
        cxy = (torch.sigmoid(t[..., :2]) + grid) * stride
        wh = torch.exp(t[..., 2:]) * anc
        box = torch.cat([cxy - wh / 2, cxy + wh / 2], -1) / S
        target = torch.as_tensor(target).to(self.dt)
        gt = target[..., 7:]
        tl, br = torch.max(box[..., :2], gt[..., :2]), torch.min(box[..., 2:], gt[..., 2:])
        area_a, area_b = torch.prod(box[..., 2:] - box[..., :2], -1), torch.prod(gt[..., 2:] - gt[..., :2], -1)
        en = (tl < br).to(self.dt).prod(-1)
        area_i = torch.prod(br - tl, -1) * en
        iou = area_i / (area_a + area_b - area_i)
        obj, gcls, gt_t, wgt = target[..., 0], target[..., 1].long(), target[..., 2:6], target[..., 6]
        mask = (obj > 0).to(self.dt)
        sg = torch.sigmoid(conf)
        gt_conf = iou.detach()
        conf_loss = torch.sum(5.0 * (obj == 1).to(self.dt) * (sg - gt_conf) ** 2 + (obj == 0).to(self.dt) * sg ** 2) / B
        cls_loss = torch.sum(F.cross_entropy(cls.permute(0, 2, 1), gcls, reduction="none") * mask) / B
        txty = torch.sum(torch.sum(F.binary_cross_entropy_with_logits(t[..., :2], gt_t[..., :2], reduction="none"), -1) * wgt * mask) / B
        twth = torch.sum(torch.sum(F.mse_loss(t[..., 2:], gt_t[..., 2:], reduction="none"), -1) * wgt * mask) / B
        iou_loss = torch.sum(F.smooth_l1_loss(iou, mask, reduction="none")) / B
        return conf_loss, cls_loss, txty + twth, iou_loss

    def train_step(self, x, target, S, lr=1e-3, momentum=0.9, weight_decay=5e-4):
        """-> (losses[4] floats, {name: grad}) ; parameters and running statistics are updated in place."""
        for v in self.params().values():
            v.grad = None
        losses = self.losses(self.forward_raw(x), target, S)
        sum(losses).backward()
        grads = {k: v.grad.clone() for k, v in self.params().items()}
        with torch.no_grad():
            for k, v in self.params().items():
                g = v.grad + weight_decay * v
                buf = self.momentum_buf.get(k)
                buf = g.clone() if buf is None else buf.mul_(momentum).add_(g)
                self.momentum_buf[k] = buf
                v.sub_(lr * buf)
        return [float(l.detach()) for l in losses], grads
