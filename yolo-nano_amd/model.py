"""Host-side mirror of the reference's ``YOLONano`` (models/yolo_nano.py:12-376).

Same constructor, attributes, helper methods, return types and ``state_dict`` key
set (469 keys) as the reference, so ``eval.py`` / ``test.py`` / ``benchmark.py``
style callers drop in; the arithmetic runs in libyolonano_hip.so through the C ABI
(include/yolonano_hip.h).  torch is plumbing here: parameter storage, device
memory, streams.  There is no CPU fallback — without the HIP library or a GPU the
model refuses to run.

Differences from the reference, on purpose:
  * ``forward_batch(x)`` finishes every image of the batch on the device; the
    reference's ``forward`` only finishes image 0 (models/yolo_nano.py:365-367) —
    ``forward`` keeps that behaviour.
  * ``backbone='0.5x'`` (and 1.5x/2.0x) are accepted; the reference prints and
    exits (models/yolo_nano.py:35-37) although its backbone supports them.
  * equal NMS scores inside a class: higher candidate index first (the reference
    inherits numpy's unstable argsort order).
"""
import numpy as np
import torch
import torch.nn as nn

from . import arch
from .capi import Handle, YnError, YnRangeError


# ---- parameter containers with the reference's module tree (keys must match exactly) -----------
class Conv(nn.Module):
    """utils/modules.py:8-18 — Conv2d(bias=True) + BatchNorm2d + LeakyReLU(0.1)."""

    def __init__(self, c1, c2, k, s=1, p=0, d=1, g=1, leaky=True):
        super().__init__()
        self.convs = nn.Sequential(
            nn.Conv2d(c1, c2, k, stride=s, padding=p, dilation=d, groups=g),
            nn.BatchNorm2d(c2),
            nn.LeakyReLU(0.1, inplace=True) if leaky else nn.Identity())

    def forward(self, x):
        raise YnError("sub-modules are parameter containers; run the model through YOLONano.forward (HIP path)")


class ShuffleV2Block(nn.Module):
    """backbone/shufflenetv2.py:31-67 (parameter layout only)."""

    def __init__(self, inp, oup, stride):
        super().__init__()
        if not (1 <= stride <= 3):
            raise ValueError("illegal stride value")
        self.stride = stride
        bf = oup // 2
        assert (stride != 1) or (inp == bf << 1)
        if stride > 1:
            self.branch1 = nn.Sequential(
                nn.Conv2d(inp, inp, 3, stride, 1, bias=False, groups=inp), nn.BatchNorm2d(inp),
                nn.Conv2d(inp, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True))
        else:
            self.branch1 = nn.Sequential()
        self.branch2 = nn.Sequential(
            nn.Conv2d(inp if stride > 1 else bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True),
            nn.Conv2d(bf, bf, 3, stride, 1, bias=False, groups=bf), nn.BatchNorm2d(bf),
            nn.Conv2d(bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True))

    def forward(self, x):
        raise YnError("sub-modules are parameter containers; run the model through YOLONano.forward (HIP path)")


class ShuffleNetV2(nn.Module):
    """backbone/shufflenetv2.py:81-129 (parameter layout only)."""

    def __init__(self, model_size="1.0x"):
        super().__init__()
        if model_size not in arch.STAGE_CH:
            raise NotImplementedError(model_size)
        self.model_size = model_size
        self.stage_repeats = list(arch.STAGE_REPEATS)
        self._stage_out_channels = [arch.STEM_CH] + list(arch.STAGE_CH[model_size])
        self.conv1 = nn.Sequential(nn.Conv2d(3, arch.STEM_CH, 3, 2, 1, bias=False), nn.BatchNorm2d(arch.STEM_CH), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        cin = arch.STEM_CH
        for name, rep, cout in zip(("stage2", "stage3", "stage4"), self.stage_repeats, arch.STAGE_CH[model_size]):
            seq = [ShuffleV2Block(cin, cout, 2)] + [ShuffleV2Block(cout, cout, 1) for _ in range(rep - 1)]
            setattr(self, name, nn.Sequential(*seq))
            cin = cout
        self._initialize_weights()

    def _initialize_weights(self):
        """backbone/shufflenetv2.py:131-154"""
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.01 if "first" in name else 1.0 / m.weight.shape[1])
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0.0001)
                nn.init.constant_(m.running_mean, 0)

    def forward(self, x):
        raise YnError("sub-modules are parameter containers; run the model through YOLONano.forward (HIP path)")


def shufflenetv2(model_size="1.0x", pretrained=False, **kwargs):
    """backbone/shufflenetv2.py:170-182.  There is no network here, so `pretrained` weights cannot
    be downloaded; load a state dict instead."""
    if pretrained:
        raise YnError("pretrained ImageNet weights need a download; load_state_dict() a checkpoint instead")
    return ShuffleNetV2(model_size=model_size)


def fuse_conv_bn(module):
    """utils/fuse_conv_bn.py:25-53 — fold each BN into the Conv2d that precedes it among its siblings and
    replace the BN by Identity.  The native path folds on the device anyway (yn_fold_bn); this keeps the
    reference's API and produces the same fused state dict."""
    last_conv, last_name = None, None
    for name, child in module.named_children():
        if isinstance(child, (nn.modules.batchnorm._BatchNorm, nn.SyncBatchNorm)):
            if last_conv is None:
                continue
            with torch.no_grad():
                w = last_conv.weight
                b = last_conv.bias if last_conv.bias is not None else torch.zeros_like(child.running_mean)
                f = child.weight / torch.sqrt(child.running_var + child.eps)
                last_conv.weight = nn.Parameter(w * f.reshape([last_conv.out_channels, 1, 1, 1]))
                last_conv.bias = nn.Parameter((b - child.running_mean) * f + child.bias)
            module._modules[last_name] = last_conv
            module._modules[name] = nn.Identity()
            last_conv = None
        elif isinstance(child, nn.Conv2d):
            last_conv, last_name = child, name
        else:
            fuse_conv_bn(child)
    return module


class YOLONano(nn.Module):
    def __init__(self, device, input_size=None, num_classes=20, trainable=False, conf_thresh=0.001, nms_thresh=0.50,
                 anchor_size=None, backbone="1.0x", diou_nms=False):
        super().__init__()
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.input_size = input_size
        self.num_classes = num_classes
        self.trainable = trainable
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        self.use_diou_nms = bool(diou_nms)
        self.nms_processor = self.diou_nms if diou_nms else self.nms
        self.bk = backbone
        self.stride = [8, 16, 32]
        self.anchor_list = [[float(a), float(b)] for a, b in anchor_size]
        self.anchor_size = torch.tensor(anchor_size).view(3, len(anchor_size) // 3, 2)
        self.num_anchors = self.anchor_size.size(1)
        if backbone not in arch.STAGE_CH:
            raise YnError("unknown backbone %r; supported: %s" % (backbone, sorted(arch.STAGE_CH)))
        self.grid_cell, self.stride_tensor, self.all_anchors_wh = self.create_grid(input_size)

        self.backbone = shufflenetv2(model_size=backbone, pretrained=False)
        c3, c4, c5 = arch.STAGE_CH[backbone]
        self.conv1x1_0 = Conv(c3, 96, k=1)
        self.conv1x1_1 = Conv(c4, 96, k=1)
        self.conv1x1_2 = Conv(c5, 96, k=1)
        self.smooth_0 = Conv(96, 96, k=3, p=1)
        self.smooth_1 = Conv(96, 96, k=3, p=1)
        self.smooth_2 = Conv(96, 96, k=3, p=1)
        self.smooth_3 = Conv(96, 96, k=3, p=1)
        hc = self.num_anchors * (1 + self.num_classes + 4)
        for i in (1, 2, 3):
            setattr(self, "head_det_%d" % i, nn.Sequential(
                Conv(96, 96, k=3, p=1, g=96), Conv(96, 96, k=1), Conv(96, 96, k=3, p=1, g=96), Conv(96, 96, k=1),
                nn.Conv2d(96, hc, 1)))
        if self.trainable:
            self.init_bias()
        self._handle = None
        self._handle_keys = None
        self._sig = None
        self._graph = False
        self._bound = None                   # the handle whose flat training buffers hold the parameters
        self._stats_stale = False            # module BN buffers newer than the handle's
        self._stats_dirty = False            # handle BN statistics newer than the module's
        self.dp_average = True               # data-parallel: average gradients over ranks inside backward()
        self._train_dtype = "f32"            # arithmetic of the training step (train_precision)
        self._exact_f32 = False              # set once an activation left the split-f16 range (_range_guarded): f32-MFMA kernels from then on

    def __deepcopy__(self, memo):
        """deepcopy (utils/misc.py:70 ModelEMA) must not clone the native handle; the copy builds its own."""
        import copy
        self.sync_running_stats()
        saved = (self._handle, self._handle_keys, self._sig, self._bound)
        self._handle, self._handle_keys, self._sig, self._bound = None, None, None, None
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                setattr(new, k, copy.deepcopy(v, memo))
        finally:
            self._handle, self._handle_keys, self._sig, self._bound = saved
        return new

    # ---- reference helpers ------------------------------------------------------------------------
    def init_bias(self):
        """models/yolo_nano.py:77-83"""
        bias_value = -torch.log(torch.tensor((1. - 0.01) / 0.01))
        with torch.no_grad():
            for i in (1, 2, 3):
                getattr(self, "head_det_%d" % i)[-1].bias[..., :self.num_anchors].fill_(float(bias_value))

    def create_grid(self, input_size):
        """models/yolo_nano.py:86-112 -> (grid [1,HW,1,2], stride [1,HW,A,2], anchors [1,HW,A,2])"""
        A = self.num_anchors
        g, st, aw = [], [], []
        for ind, s in enumerate(self.stride):
            ws = hs = input_size // s
            gy, gx = torch.meshgrid(torch.arange(hs), torch.arange(ws), indexing="ij")
            g.append(torch.stack([gx, gy], dim=-1).float().view(1, hs * ws, 1, 2))
            st.append(torch.ones([1, hs * ws, A, 2]) * s)
            aw.append(self.anchor_size[ind].repeat(hs * ws, 1, 1))
        dev = self.device if (self.device.type != "cuda" or torch.cuda.is_available()) else torch.device("cpu")
        return (torch.cat(g, dim=1).to(dev), torch.cat(st, dim=1).to(dev), torch.cat(aw, dim=0).to(dev).unsqueeze(0))

    def set_grid(self, input_size):
        """models/yolo_nano.py:115-117"""
        self.input_size = input_size
        self.grid_cell, self.stride_tensor, self.all_anchors_wh = self.create_grid(input_size)
        if self._handle is not None:
            self._handle.set_grid(input_size)

    # ---- native handle ------------------------------------------------------------------------------
    def use_graph(self, on=True):
        """hipGraph-capture the fixed-shape inference pipeline (BASELINE config 5)."""
        self._graph = bool(on)
        if self._handle is not None:
            self._handle.use_graph(on)

    def _state_signature(self, sd):
        return tuple((k, v.data_ptr(), v._version, tuple(v.shape)) for k, v in sd.items())

    def handle(self, batch=1):
        """The yn_handle with the module's current weights loaded and folded."""
        sd = self.state_dict()
        first = next(iter(sd.values()))
        if not first.is_cuda:
            raise YnError("YOLONano parameters are on %s: the HIP path needs them on the GPU (model.to('cuda')); "
                          "there is no CPU fallback" % first.device)
        sig = self._state_signature(sd)
        keys = tuple(sd.keys())
        if self._handle is None or self._handle_keys != keys or self._handle.device != first.device:
            if self._handle is not None:
                self._handle.close()
            self._handle = Handle(self.input_size, self.num_classes, self.anchor_list, self.bk, self.conf_thresh, self.nms_thresh,
                                  self.use_diou_nms, max_batch=batch, device=first.device)
            self._handle_keys = keys
            self._handle.use_graph(self._graph)
            if self._exact_f32:
                self._handle.exact_f32(True)
            self._sig = None
        h = self._handle
        h.follow_current_stream()
        if h.S != self.input_size:
            h.set_grid(self.input_size)
        h.set_thresholds(self.conf_thresh, self.nms_thresh, self.use_diou_nms)
        if sig != self._sig:
            for k, v in sd.items():
                if not k.endswith("num_batches_tracked"):
                    h.load_param(k, v)
            h.fold_bn()
            self._sig = sig
        return h

    # ---- decode / NMS helpers with the reference's signatures -----------------------------------------
    def decode_xywh(self, txtytwth_pred):
        """models/yolo_nano.py:120-136 (derived from the native xyxy decode)."""
        xyxy = self.decode_boxes(txtytwth_pred)
        cxy = (xyxy[..., :2] + xyxy[..., 2:]) / 2
        wh = xyxy[..., 2:] - xyxy[..., :2]
        return torch.cat([cxy, wh], -1)

    def decode_boxes(self, txtytwth_pred):
        """models/yolo_nano.py:139-156 : [B, HW, A, 4] -> [B, HW*A, 4] xyxy in pixels (yn_decode_boxes)."""
        t = torch.as_tensor(txtytwth_pred)
        h = self.handle()
        return h.decode_boxes(t.to(h.device))

    def nms(self, dets, scores):
        """models/yolo_nano.py:159-188 : numpy [n,4], [n] -> list of kept indices in pick order (yn_nms)."""
        return self._nms(dets, scores, False)

    def diou_nms(self, dets, scores):
        """models/yolo_nano.py:191-242"""
        return self._nms(dets, scores, True)

    def _nms(self, dets, scores, diou):
        h = self.handle()
        d = torch.as_tensor(np.ascontiguousarray(dets, dtype=np.float32)).to(h.device)
        s = torch.as_tensor(np.ascontiguousarray(scores, dtype=np.float32)).to(h.device)
        return h.nms(d, s, self.nms_thresh, diou).cpu().numpy().astype(np.int64).tolist()

    def postprocess(self, all_local, all_conf):
        """models/yolo_nano.py:245-279 : numpy [N,4], [N,C] -> (bboxes [K,4] f32, scores [K] f32, cls_inds [K] i64)."""
        h = self.handle()
        b = torch.as_tensor(np.ascontiguousarray(all_local, dtype=np.float32)).to(h.device)[None]
        c = torch.as_tensor(np.ascontiguousarray(all_conf, dtype=np.float32)).to(h.device)[None]
        out = h.postprocess(b, c)
        return self._to_host(out, 0)

    @staticmethod
    def _to_host(out, b, k=None):
        boxes, scores, cls, _, count = out
        k = int(count[b].item()) if k is None else k
        if k < 0:                                              # yn_infer's range mark (compact_kernel): the batch is invalid
            raise YnRangeError("yn_infer: an activation exceeded the split-f16 range (|x| >= 65504)")
        # fresh, writable, caller-owned arrays: callers rescale them in place (benchmark.py:69-71)
        return (boxes[b, :k].cpu().numpy().copy(), scores[b, :k].cpu().numpy().copy(),
                cls[b, :k].cpu().numpy().astype(np.int64))

    # ---- forward ------------------------------------------------------------------------------------------
    def _range_guarded(self, h, run):
        """Run `run(h)` on the default (split-f16) family and check the handle's range guard (yn_range_status): if an activation
        reached 65504 — outside x = hi + lo * 2^-11, where the reference's fp32 is still finite — switch the handle to the f32-MFMA
        family for good and run again.  Only forward_raw (device tensors out, nothing read back) goes through this form and pays its
        synchronising 4-byte read-back; forward / forward_batch learn the same from yn_infer's counts (_infer_guarded)."""
        out = run(h)
        if not self._exact_f32:
            _, overflow = h.range_status()
            if overflow:
                import warnings
                warnings.warn("yolo_nano_amd: an activation exceeded the split-f16 range (|x| >= 65504); re-running this and all later "
                              "forwards of the model on the exact f32-MFMA kernels (yn_exact_f32)")
                self._exact_f32 = True
                h.exact_f32(True)
                out = run(h)
        return out

    def _infer_guarded(self, h, xf, finish):
        """yn_infer + `finish(out)` (the read-back the caller makes anyway).  yn_infer itself reports an activation outside the split-f16 range
        through NEGATIVE counts (compact_kernel), so the guard costs no extra synchronisation here: on that mark the handle is switched to the
        f32-MFMA family for good and the call runs again."""
        try:
            return finish(h.infer(xf))
        except YnRangeError:
            if self._exact_f32:
                raise
            import warnings
            warnings.warn("yolo_nano_amd: an activation exceeded the split-f16 range (|x| >= 65504); re-running this and all later "
                          "forwards of the model on the exact f32-MFMA kernels (yn_exact_f32)")
            h.range_status()                                   # clears the device flag
            self._exact_f32 = True
            h.exact_f32(True)
            return finish(h.infer(xf))

    def forward_raw(self, x):
        """Raw head tensors as the reference's hooks see them: three NCHW views [B, A(5+C), H, W]."""
        h = self.handle(x.shape[0])
        xf = x.float()
        return [t.permute(0, 3, 1, 2) for t in self._range_guarded(h, lambda hh: hh.forward_raw(xf))]

    def forward_batch(self, x):
        """Eval-mode forward for EVERY image: list of (bboxes, scores, cls_inds) numpy triples."""
        h = self.handle(x.shape[0])
        xf = x.float()
        return self._infer_guarded(h, xf, h.detections_to_host)      # two device-to-host copies per batch (yn_pack_detections)

    # ---- training (models/yolo_nano.py:332-358, train.py:219-231) -----------------------------------------
    def _train_handle(self, batch):
        """The handle with this module's parameters living INSIDE its flat training buffers: every nn.Parameter becomes a
        view of `flat_params` (named_parameters() order == the C ABI's flat order), so the HIP step, torch.optim.SGD,
        `yolo_nano_amd.SGD`, load_state_dict and ModelEMA all see the same storage and nothing is copied per step."""
        h = self._handle
        if h is None or self._bound is not h:
            h = self.handle(batch)                           # loads the current weights + BN statistics
            n = h.train_bind()
            off = 0
            for name, p in self.named_parameters():
                sl = h.param_slice(name)
                if sl.start != off or sl.stop - sl.start != p.numel():
                    raise YnError("parameter order mismatch at %s" % name)
                off = sl.stop
                p.data = h.flat_params[sl].view(p.shape)
                p.grad = None
            if off != n:
                raise YnError("flat parameter buffer holds %d values, the module %d" % (n, off))
            self._bound = h
            self._stats_stale = False
            h.train_precision(self._train_dtype)
        h.follow_current_stream()
        if h.S != self.input_size:                           # multi-scale training: train.py:202-208
            h.set_grid(self.input_size)
        if self._stats_stale:                                # load_state_dict after binding: push the BN statistics down
            for k, v in self.state_dict().items():
                if k.endswith(("running_mean", "running_var")):
                    h.load_param(k, v)
            self._stats_stale = False
        return h

    def train_precision(self, dtype="f32"):
        """Arithmetic of the training step: "f32" (what train.py runs) or "f16" (BASELINE configs[2]: fp16 activation / gradient
        storage and f16 MFMA, fp32 master weights, loss scale on the device - yn_train_precision). Returns self."""
        if dtype not in ("f32", "fp32", "f16", "fp16"):
            raise YnError("train_precision: %r is not one of 'f32', 'f16'" % (dtype,))
        self._train_dtype = "f16" if dtype in ("f16", "fp16") else "f32"
        if self._bound is not None:
            self._bound.train_precision(self._train_dtype)
        return self

    def make_targets(self, label_lists):
        """tools.multi_gt_creator for this model's input size / anchors, on the model's device -> [B, N, 11] CUDA tensor."""
        h = self._handle if self._handle is not None else self.handle(len(label_lists))
        if h.S != self.input_size:
            h.set_grid(self.input_size)
        return h.make_targets(label_lists, self.anchor_list)

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._sig = None
        self._stats_stale = True
        return out

    def sync_running_stats(self):
        """Bring the BatchNorm running statistics the HIP training step updates back into the module's buffers
        (state_dict() / checkpoints / the eval-mode fold read them from there)."""
        h = self._handle
        if h is None or self._bound is not h or not self._stats_dirty:
            return
        for k, v in self.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                v.copy_(torch.as_tensor(h.read_param(k, tuple(v.shape))))
        self._stats_dirty = False

    def state_dict(self, *a, **kw):
        if getattr(self, "_stats_dirty", False) and not getattr(self, "_in_sync", False):
            self._in_sync = True
            try:
                self.sync_running_stats()
            finally:
                self._in_sync = False
        return super().state_dict(*a, **kw)

    def forward(self, x, target=None):
        if self.trainable:
            if target is None:
                raise YnError("trainable forward needs target [B, N, 11] (tools.multi_gt_creator layout)")
            return _TrainStep.apply(self, x, target, *list(self.parameters()))
        h = self.handle(x.shape[0])
        xf = x.float()
        return self._infer_guarded(h, xf, lambda out: self._to_host(out, 0))          # batch element 0 only, as models/yolo_nano.py:365-367


class ModelEMA(object):
    """utils/misc.py:67-86 with the same constructor / attributes (`ema`, `updates`, `decay`) and `update(model)`; the
    in-place lerp of every floating-point state-dict entry runs as yn_ema_update — TWO launches per update: one over the flat
    parameter buffer (model and EMA copy both bound to their training buffers) and one over the flat BatchNorm-statistics
    buffer (the 148 running_mean / running_var tensors of model and copy are re-homed into one buffer each, same order);
    one launch per tensor only for whatever cannot be re-homed (CPU / non-fp32 entries)."""

    def __init__(self, model, decay=0.9999, updates=0):
        import copy
        import math
        m = model.module if hasattr(model, "module") else model
        self.ema = copy.deepcopy(m).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000.))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        # deepcopy gives every tensor its own storage: move the copy's parameters / its float buffers into one flat buffer each
        # (named_parameters / buffers order, the layout of the model's training buffers) so that update() is a single launch each
        self._rehome(list(self.ema.parameters()))
        self._rehome(self._float_buffers(self.ema))

    @staticmethod
    def _float_buffers(module):
        return [b for b in module.buffers() if b.dtype.is_floating_point]

    @staticmethod
    def _rehome(tensors):
        """Make every tensor of the list a view of ONE new flat float32 buffer (values kept); None when they cannot be."""
        if not tensors or not all(t.is_cuda and t.dtype == torch.float32 for t in tensors):
            return None
        flat = torch.empty(sum(t.numel() for t in tensors), dtype=torch.float32, device=tensors[0].device)
        off = 0
        for t in tensors:
            flat[off:off + t.numel()].copy_(t.data.reshape(-1))
            t.data = flat[off:off + t.numel()].view(t.shape)
            off += t.numel()
        return flat

    @staticmethod
    def _flat_view(tensors):
        """The single contiguous buffer the tensors are consecutive views of (after _rehome / YOLONano._train_handle / a deepcopy
        of either), else None."""
        if not tensors or any(not t.is_cuda or t.dtype != torch.float32 for t in tensors):
            return None
        st = tensors[0].data.untyped_storage()
        off = tensors[0].data.storage_offset()
        base = off
        for t in tensors:
            if t.data.untyped_storage().data_ptr() != st.data_ptr() or t.data.storage_offset() != off or not t.data.is_contiguous():
                return None
            off += t.numel()
        return torch.empty(0, dtype=torch.float32, device=tensors[0].device).set_(st, base, (off - base,))

    @staticmethod
    def _flat_of(module):
        return ModelEMA._flat_view(list(module.parameters()))

    def update(self, model):
        m = model.module if hasattr(model, "module") else model
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            h = m.handle() if getattr(m, "_handle", None) is None else m._handle
            msd, esd = m.state_dict(), self.ema.state_dict()           # (state_dict() first: it syncs the handle's running statistics in place)
            done = set()
            fm, fe = self._flat_of(m), self._flat_of(self.ema)
            if fm is not None and fe is not None and fm.numel() == fe.numel():
                h.ema_update(fe, fm, d)
                done = {k for k, _ in m.named_parameters()}
            # the BatchNorm statistics: one launch over the two flat buffers (the model's are re-homed on the first update)
            mb, eb = self._float_buffers(m), self._float_buffers(self.ema)
            if len(mb) == len(eb) and all(a.shape == b.shape for a, b in zip(mb, eb)):
                fmb = self._flat_view(mb)
                if fmb is None:
                    fmb = self._rehome(mb)
                feb = self._flat_view(eb)
                if feb is None:
                    feb = self._rehome(eb)
                if fmb is not None and feb is not None and fmb.numel() == feb.numel():
                    h.ema_update(feb, fmb, d)
                    done |= {k for k, b in m.named_buffers() if b.dtype.is_floating_point}
            for k, v in esd.items():
                if k in done or not v.dtype.is_floating_point:
                    continue
                src = msd[k].detach()
                if v.is_contiguous() and src.is_contiguous() and v.dtype == torch.float32 and v.is_cuda:
                    h.ema_update(v, src, d)
                else:
                    v *= d
                    v += (1. - d) * src
        # yn_ema_update writes through data_ptr(): torch's version counters do not move, so the copy's cached handle would keep the
        # weights it folded before this update (state_dict() right, inference stale) - drop the signature, the next eval re-loads
        self.ema._sig = None


class ValTransforms(object):
    """data/transforms.py:445-458 (Resize :73-119 -> Normalize :59-70 -> ToTensor :394-398) with the pixel work on the device.

        x, boxes, labels, scale, offset = ValTransforms(size)(image, boxes, labels)

    Same constructor, call signature and return tuple as the reference; `image` is the uint8 HxWx3 BGR array cv2.imread gives.
    The image travels to the GPU as uint8 (3 bytes per pixel instead of the 12 of the float tensor the reference uploads) and
    yn_preprocess writes the normalised letterboxed RGB CHW float32 tensor (cv2.resize's 8-bit INTER_LINEAR arithmetic,
    restated: oracle/preprocess.py explains why that parity is unpinned).  Resize's integer geometry, `scale` and `offset` are
    computed here with the reference's own expressions.  `out=` writes straight into one [3,size,size] slot of a batch."""

    def __init__(self, size=640, mean=(0.406, 0.456, 0.485), std=(0.225, 0.224, 0.229), handle=None, device=None):
        self.size = size
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self._handle = handle
        self._device = device

    def _h(self):
        if self._handle is None:                               # a bare handle: only its stream / error plumbing is used
            from . import capi
            from . import arch
            dev = self._device if self._device is not None else torch.device("cuda", torch.cuda.current_device())
            self._handle = capi.Handle(32, 1, arch.MULTI_ANCHOR_SIZE, "1.0x", device=dev)
        return self._handle

    def geometry(self, h0, w0):
        """Resize.__call__ (data/transforms.py:79-116): (rw, rh, left, top, scale, offset)."""
        size = self.size
        if h0 > w0:
            r = w0 / h0
            w, h = int(r * size), size
            left = (h - w) // 2
            return w, h, left, 0, np.array([[w / h, 1., w / h, 1.]]), np.array([[left / h, 0., left / h, 0.]])
        if h0 < w0:
            r = h0 / w0
            w, h = size, int(r * size)
            top = (w - h) // 2
            return w, h, 0, top, np.array([1., h / w, 1., h / w]), np.array([[0., top / w, 0., top / w]])
        return size, size, 0, 0, 1., np.zeros([1, 4])

    def __call__(self, image, boxes=None, labels=None, scale=None, offset=None, out=None):
        h0, w0, _ = image.shape
        rw, rh, left, top, scale, offset = self.geometry(h0, w0)
        hd = self._h()
        img = torch.as_tensor(np.ascontiguousarray(image, dtype=np.uint8)).to(hd.device, non_blocking=True)
        x = hd.preprocess(img, rw, rh, left, top, self.size, self.mean, self.std, out=out)
        if boxes is not None:
            boxes = boxes * scale + offset
        return x, boxes, labels, scale, offset


    def batch(self, images, out=None):
        """A list of uint8 HxWx3 BGR arrays (any sizes) -> (x float32 [n,3,size,size] on the device, scales, offsets): the loop
        of benchmark.py:58 / vocapi_evaluator.py:64 over a batch, one kernel launch per 32 images."""
        hd = self._h()
        dev_imgs, geoms, scales, offsets = [], [], [], []
        for im in images:
            rw, rh, left, top, scale, offset = self.geometry(im.shape[0], im.shape[1])
            dev_imgs.append(torch.as_tensor(np.ascontiguousarray(im, dtype=np.uint8)).to(hd.device, non_blocking=True))
            geoms.append((rw, rh, left, top)); scales.append(scale); offsets.append(offset)
        return hd.preprocess_batch(dev_imgs, geoms, self.size, self.mean, self.std, out=out), scales, offsets


def rescale_boxes(bboxes, scale, offset, size):
    """benchmark.py:66-69 / evaluator/vocapi_evaluator.py:72-75: detections of the letterboxed square back to pixels of the
    original image (`size` = np.array([[w, h, w, h]])); in place, like the reference."""
    bboxes -= offset
    bboxes /= scale
    bboxes *= size
    return bboxes


class TestTimeAugmentation(object):
    """utils/misc.py:90-148: multi-scale (scale_range) x horizontal flip forwards of image 0, merged by per-class NMS.
    Same constructor and call signature; the forwards run through the model's handle, the merge through yn_nms_merge."""
    __test__ = False                                           # not a pytest class

    def __init__(self, num_classes=80, nms_thresh=0.4, scale_range=[320, 640, 32]):
        self.num_classes = num_classes
        self.nms_thresh = nms_thresh
        self.scales = np.arange(scale_range[0], scale_range[1] + 1, scale_range[2])

    def __call__(self, x, model):
        bboxes_list, scores_list, labels_list = [], [], []
        size0 = model.input_size
        for s in self.scales:
            s = int(s)
            if x.size(-1) == s and x.size(-2) == s:
                x_scale = x
            else:
                x_scale = torch.nn.functional.interpolate(input=x, size=(s, s), mode='bilinear', align_corners=False)
            model.set_grid(s)
            bboxes, scores, labels = model(x_scale)
            bboxes_list.append(bboxes); scores_list.append(scores); labels_list.append(labels)
            bboxes, scores, labels = model(torch.flip(x_scale, [-1]))
            bboxes = bboxes.copy()
            bboxes[:, 0::2] = 1.0 - bboxes[:, 2::-2]           # utils/misc.py:126
            bboxes_list.append(bboxes); scores_list.append(scores); labels_list.append(labels)
        if size0 is not None:
            model.set_grid(size0)
        bboxes = np.concatenate(bboxes_list)
        scores = np.concatenate(scores_list)
        labels = np.concatenate(labels_list)
        if len(bboxes) == 0:
            return bboxes, scores, labels
        h = model.handle()
        dev = h.device
        ob, osc, oc, _ = h.nms_merge(torch.as_tensor(bboxes).to(dev), torch.as_tensor(scores).to(dev),
                                     torch.as_tensor(labels.astype(np.int32)).to(dev), self.num_classes, self.nms_thresh)
        return ob.cpu().numpy().copy(), osc.cpu().numpy().copy(), oc.cpu().numpy().astype(np.int64)


_target_handles = {}


def multi_gt_creator(input_size, strides, label_lists, anchor_size, device=None):
    """tools.multi_gt_creator (tools.py:97-216) with the reference's signature, computed on the GPU (yn_make_targets):
    -> float32 CUDA tensor [B, N, 11] (the reference returns a CPU tensor that train.py:216 moves to the device)."""
    if list(strides) != [8, 16, 32]:
        raise YnError("strides must be [8, 16, 32] (models/yolo_nano.py:23), got %r" % (strides,))
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (int(input_size), str(dev))
    h = _target_handles.get(key)
    if h is None:
        if len(_target_handles) >= 16:                       # multi-scale training cycles through 10 sizes (train.py:202-208)
            _target_handles.pop(next(iter(_target_handles))).close()
        h = _target_handles[key] = Handle(int(input_size), 1, [list(a) for a in anchor_size], "1.0x", max_batch=1, device=dev)
    return h.make_targets(label_lists, anchor_size)


class _TrainStep(torch.autograd.Function):
    """`model(images, target=targets)` in train mode: yn_train_step(do_update=0) runs forward + loss + backward in one
    call and leaves d(conf+cls+bbox+iou)/d(parameters) in the flat gradient buffer; autograd's backward() only hands
    views of that buffer to the parameters (after the data-parallel all-reduce), so train.py:219-231 runs unchanged:
        conf_loss, cls_loss, bbox_loss, iou_loss = model(images, target=targets)
        (conf_loss + cls_loss + bbox_loss + iou_loss).backward(); optimizer.step(); optimizer.zero_grad()"""

    @staticmethod
    def forward(ctx, model, x, target, *params):
        h = model._train_handle(x.shape[0])
        losses = h.train_step(x, target, update=False)
        model._stats_dirty = True
        model._sig = None
        ctx.model = model
        return tuple(losses.unbind(0))

    @staticmethod
    def backward(ctx, *g):
        model = ctx.model
        h = model._handle
        gs = torch.stack([t if t is not None else torch.zeros_like(g[0]) for t in g]).tolist()
        scale = float(gs[0])
        if any(v != scale for v in gs):
            raise YnError("the four losses must enter the total with equal weights (train.py:222): the HIP backward "
                          "differentiates their sum")
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(h.flat_grads, op=dist.ReduceOp.SUM)          # ONE flat bucket over RCCL / xGMI
            if model.dp_average:
                scale /= dist.get_world_size()
        if scale != 1.0:
            h.flat_grads.mul_(scale)
        out = [None, None, None]
        off = 0
        for p in model.parameters():
            out.append(h.flat_grads[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return tuple(out)


class SGD:
    """torch.optim.SGD(lr, momentum, weight_decay) as train.py:167-171 builds it, as ONE fused HIP kernel over the model's
    flat parameter / gradient / momentum buffers (yn_sgd_step).  `param_groups[0]['lr']` is what train.py's set_lr writes."""

    def __init__(self, model, lr=1e-3, momentum=0.9, weight_decay=5e-4):
        self.model = model
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay, "params": list(model.parameters())}]

    def zero_grad(self, set_to_none=True):
        for p in self.param_groups[0]["params"]:
            p.grad = None

    def step(self):
        m, g = self.model, self.param_groups[0]
        h = m._handle
        if h is None or m._bound is not h:
            raise YnError("SGD.step() before the first training forward/backward")
        h.sgd_step(h.flat_params, h.flat_grads, h.flat_momentum, g["lr"], g["momentum"], g["weight_decay"])
        m._sig = None
