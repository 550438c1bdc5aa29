"""ctypes binding of libyolonano_hip.so (include/yolonano_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or no GPU is
visible, creating a handle raises.  torch is imported first on purpose — it loads
its bundled libamdhip64.so.7, and the library then resolves the same runtime by
soname, so torch tensors' ``data_ptr()`` and torch streams are valid in it.
"""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime)

from . import arch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyolonano_hip.so")

_vp = ctypes.c_void_p
_i32, _f32 = ctypes.c_int, ctypes.c_float
_i64p = ctypes.POINTER(ctypes.c_int64)

BACKBONE_ID = {"0.5x": 0, "1.0x": 1, "1.5x": 2, "2.0x": 3}


class YnConfig(ctypes.Structure):
    _fields_ = [("input_size", _i32), ("num_classes", _i32), ("num_anchors", _i32),
                ("anchors", _f32 * 18), ("backbone", _i32), ("conf_thresh", _f32), ("nms_thresh", _f32),
                ("diou_nms", _i32), ("max_batch", _i32), ("device", _i32), ("stream", _vp)]


# name -> (restype, argtypes); every symbol include/yolonano_hip.h declares
SIGNATURES = {
    "yn_abi_version": (_i32, []),
    "yn_create": (_i32, [ctypes.POINTER(YnConfig), ctypes.POINTER(_vp)]),
    "yn_destroy": (None, [_vp]),
    "yn_last_error": (ctypes.c_char_p, [_vp]),
    "yn_set_grid": (_i32, [_vp, _i32]),
    "yn_set_stream": (_i32, [_vp, _vp]),
    "yn_set_thresholds": (_i32, [_vp, _f32, _f32, _i32]),
    "yn_num_predictions": (_i32, [_vp]),
    "yn_use_graph": (_i32, [_vp, _i32]),
    "yn_synchronize": (_i32, [_vp]),
    "yn_autotune": (_i32, [_vp, _i32]),
    "yn_set_pw_config": (_i32, [_vp, _i32]),
    "yn_tune_save": (_i32, [ctypes.c_char_p, _i32]),
    "yn_tune_load": (_i32, [ctypes.c_char_p, _i32]),
    "yn_allreduce_grads": (_i32, [_vp, _vp]),
    "yn_pw_config_count": (_i32, []),
    "yn_pw_f32_config_count": (_i32, []),
    "yn_unit_chain": (_i32, [_vp, _i32]),
    "yn_chain_pipe": (_i32, [_vp, _i32]),
    "yn_stage_fuse": (_i32, [_vp, _i32, _i32]),
    "yn_pw_pipe": (_i32, [_vp, _i32]),
    "yn_multi_stream": (_i32, [_vp, _i32]),
    "yn_exact_f32": (_i32, [_vp, _i32]),
    "yn_range_status": (_i32, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "yn_fuse_decode": (_i32, [_vp, _i32]),
    "yn_group_launch": (_i32, [_vp, _i32]),
    "yn_down_fuse": (_i32, [_vp, _i32]),
    "yn_tail_fuse": (_i32, [_vp, _i32]),
    "yn_nms_prefilter": (_i32, [_vp, _i32]),
    "yn_nms_sweep": (_i32, [_vp, _i32]),
    "yn_nms_sweep_segments": (_i32, [_vp, _i32, _i32]),
    "yn_load_param": (_i32, [_vp, ctypes.c_char_p, _vp, _i64p, _i32]),
    "yn_load_param_dev": (_i32, [_vp, ctypes.c_char_p, _vp, _i64p, _i32]),
    "yn_fold_bn": (_i32, [_vp]),
    "yn_get_folded": (_i32, [_vp, ctypes.c_char_p, _vp, _vp]),
    "yn_forward_raw": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "yn_forward_taps": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "yn_score_full": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "yn_decode_boxes": (_i32, [_vp, _vp, _i32, _vp]),
    "yn_create_grid": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "yn_nms": (_i32, [_vp, _vp, _vp, _i32, _f32, _i32, _vp, _vp]),
    "yn_postprocess": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "yn_infer": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "yn_pack_detections": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "yn_loss": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "yn_loss_heads": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "yn_train_param_count": (ctypes.c_int64, [_vp]),
    "yn_train_param_offset": (_i32, [_vp, ctypes.c_char_p, _i64p, _i64p]),
    "yn_train_bind": (_i32, [_vp, _vp, _vp, _vp, ctypes.c_int64]),
    "yn_train_step": (_i32, [_vp, _vp, _vp, _i32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "yn_read_param": (_i32, [_vp, ctypes.c_char_p, _vp, ctypes.c_int64]),
    "yn_train_forward": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "yn_train_skipped_steps": (_i32, [_vp, _i64p]),
    "yn_train_head_fork": (_i32, [_vp, _i32, ctypes.POINTER(ctypes.c_int)]),
    "yn_train_precision": (_i32, [_vp, _i32]),
    "yn_train_graph": (_i32, [_vp, _i32, _vp]),
    "yn_train_get_loss_scale": (_i32, [_vp, ctypes.POINTER(_f32), ctypes.POINTER(_f32)]),
    "yn_train_set_loss_scale": (_i32, [_vp, _f32, _f32]),
    "yn_make_targets": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "yn_preprocess": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "yn_preprocess_batch": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "yn_nms_merge": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "yn_ema_update": (_i32, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_double]),
    "yn_sgd_step": (_i32, [_vp, _vp, _vp, _vp, ctypes.c_int64, _f32, _f32, _f32, _f32, _i32]),
    "yn_op_dwconv3x3": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "yn_op_pwconv": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "yn_op_pwconv_shuffle": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "yn_op_conv3x3": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "yn_op_stem": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "yn_op_maxpool3x3s2": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "yn_op_shuffle_block": (_i32, [_vp, ctypes.c_char_p, _vp, _i32, _i32, _i32, _vp]),
    "yn_op_nchw_to_nhwc": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "yn_op_nhwc_to_nchw": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "yn_op_h16_conv": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "yn_op_h16_bn": (_i32, [_vp, _vp, _vp, ctypes.c_int64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "yn_op_h16_gemm_stats": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "yn_op_h16_bn_unit": (_i32, [_vp, _vp, _vp, _vp, ctypes.c_int64, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "yn_profile_enable": (_i32, [_vp, _i32]),
    "yn_profile_count": (_i32, [_vp]),
    "yn_profile_get": (_i32, [_vp, _i32, ctypes.c_char_p, _i32, ctypes.c_char_p, _i32, ctypes.POINTER(_f32),
                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


class YnError(RuntimeError):
    pass


class YnRangeError(YnError):
    """yn_infer reported (negative counts / offsets[B]) that an activation left the split-f16 range: the detections of that call are not
    valid.  Call range_status() (clears the flag), switch the handle to exact_f32(True) and run again - YOLONano does this by itself."""


def load_library():
    """dlopen the in-tree HIP library and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YnError("%s is missing — build it with `python -m yolo_nano_amd.build` "
                          "(there is no CPU fallback for the product path)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)              # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous(), "tensor must be contiguous"
        return t.data_ptr()
    return t


def tune_save(path, device=0):
    """Write the process-wide autotune table of `device` to `path` (yn_tune_save)."""
    if load_library().yn_tune_save(str(path).encode(), int(device)):
        raise YnError("yn_tune_save(%s) failed" % path)


def tune_load(path, device=0):
    """Adopt the entries of `path` for `device` (yn_tune_load) -> number adopted, -1 if unreadable."""
    return int(load_library().yn_tune_load(str(path).encode(), int(device)))


class Handle:
    """One yn_handle = one device + one stream.  Thin, 1:1 with the C ABI."""

    def __init__(self, input_size, num_classes, anchor_size, backbone="1.0x", conf_thresh=0.001,
                 nms_thresh=0.5, diou_nms=False, max_batch=1, device=None, stream=None):
        self.lib = load_library()
        if backbone not in BACKBONE_ID:
            raise YnError("unknown backbone %r" % (backbone,))
        if not torch.cuda.is_available():
            raise YnError("no MI355X visible: the YOLO-Nano HIP path needs a GPU (no CPU fallback)")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise YnError("device must be a GPU, got %s" % (dev,))
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        flat = [float(v) for wh in anchor_size for v in wh]
        if len(flat) % 6:
            raise YnError("anchor_size must hold 3*A [w,h] pairs")
        self.A = len(flat) // 6
        cfg = YnConfig()
        cfg.input_size, cfg.num_classes, cfg.num_anchors = int(input_size), int(num_classes), self.A
        for i, v in enumerate(flat):
            cfg.anchors[i] = v
        cfg.backbone = BACKBONE_ID[backbone]
        cfg.conf_thresh, cfg.nms_thresh, cfg.diou_nms = float(conf_thresh), float(nms_thresh), int(bool(diou_nms))
        cfg.max_batch, cfg.device = int(max_batch), self.device.index
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device) if stream is None else stream
            cfg.stream = st.cuda_stream
            h = _vp()
            if self.lib.yn_create(ctypes.byref(cfg), ctypes.byref(h)):
                raise YnError("yn_create: " + self.lib.yn_last_error(None).decode())
        self.h = h
        self._stream_ptr = cfg.stream
        self.C, self.S = int(num_classes), int(input_size)
        self.backbone = backbone
        self.head_ch = arch.head_channels(self.C, self.A)

    def close(self):
        if getattr(self, "h", None):
            self.lib.yn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _torch_stream(self):
        """The stream the handle launches on, as a torch stream object (for record_stream)."""
        if getattr(self, "_ts_ptr", None) != self._stream_ptr or getattr(self, "_ts", None) is None:
            self._ts = torch.cuda.ExternalStream(self._stream_ptr, device=self.device) if self._stream_ptr else torch.cuda.default_stream(self.device)
            self._ts_ptr = self._stream_ptr
        return self._ts

    def _in(self, t, dtype=torch.float32):
        """`t` as a contiguous `dtype` tensor that is safe to hand to the library as a bare pointer.  A converted / compacted COPY would go
        back to torch's caching allocator as soon as the caller's expression ends, while the kernels that read it are still queued on
        the handle's stream (which need not be torch's current one): the copy is tied to that stream with record_stream."""
        c = t if t.dtype == dtype else t.to(dtype)
        c = c.contiguous()
        if c.is_cuda and c.data_ptr() != t.data_ptr():
            c.record_stream(self._torch_stream())
        return c

    def _ck(self, rc, what):
        if rc == 2:                                             # YN_STATUS_RANGE: an earlier yn_infer's range mark has not been acknowledged
            raise YnRangeError("%s: %s" % (what, self.lib.yn_last_error(self.h).decode()))
        if rc:
            raise YnError("%s: %s" % (what, self.lib.yn_last_error(self.h).decode()))

    # ---- configuration
    @property
    def N(self):
        return self.lib.yn_num_predictions(self.h)

    def set_grid(self, S):
        self._ck(self.lib.yn_set_grid(self.h, int(S)), "yn_set_grid")
        self.S = int(S)

    def set_stream(self, stream):
        self._ck(self.lib.yn_set_stream(self.h, stream.cuda_stream), "yn_set_stream")
        self._stream_ptr = stream.cuda_stream

    def follow_current_stream(self):
        """Re-home the handle onto torch's current stream of its device when that differs from the one it launches on
        (the old stream is drained first: the activation arena is shared).  The host shim calls this before every forward so
        that `with torch.cuda.stream(s): model(x)` orders the HIP kernels with the torch ops issued on `s`."""
        st = torch.cuda.current_stream(self.device)
        if st.cuda_stream != self._stream_ptr:
            self.set_stream(st)

    def set_thresholds(self, conf, nms, diou=False):
        self._ck(self.lib.yn_set_thresholds(self.h, float(conf), float(nms), int(bool(diou))), "yn_set_thresholds")

    def use_graph(self, on=True):
        self._ck(self.lib.yn_use_graph(self.h, int(bool(on))), "yn_use_graph")

    def autotune(self, on=True):
        self._ck(self.lib.yn_autotune(self.h, int(bool(on))), "yn_autotune")

    def set_pw_config(self, index):
        """Testing aid: pin the pointwise GEMMs to one tile configuration (index < 0: back to the autotuner)."""
        self._ck(self.lib.yn_set_pw_config(self.h, int(index)), "yn_set_pw_config")

    def unit_chain(self, mode=1):
        """One kernel per stride-1 ShuffleV2 unit: 1 = where the map is large enough (default), 0 = never, 2 = always (True = 2);
        bit-identical results either way."""
        self._ck(self.lib.yn_unit_chain(self.h, 2 if mode is True else int(mode)), "yn_unit_chain")

    def chain_pipe(self, mode=1):
        """unit_pipe_kernel (the persistent per-unit tile walk): 1 = by its size rule (default), 0 = never, 2 = also for few tiles; bit-identical."""
        self._ck(self.lib.yn_chain_pipe(self.h, 2 if mode is True else int(mode)), "yn_chain_pipe")

    def stage_fuse(self, mode=1, publish_early=False):
        """stage_pipe_kernel (all but the last stride-1 unit of a stage as ONE persistent launch): 1 = from 256 tiles (default), 0 = never,
        2 = at every size; bit-identical to the per-unit launches."""
        self._ck(self.lib.yn_stage_fuse(self.h, 2 if mode is True else int(mode), int(bool(publish_early))), "yn_stage_fuse")

    def pw_pipe(self, on=True):
        """pw_pipe_kernel among the pointwise autotune candidates (default on)."""
        self._ck(self.lib.yn_pw_pipe(self.h, int(bool(on))), "yn_pw_pipe")

    def multi_stream(self, on=True):
        """Fork independent kernel chains of one forward onto the handle's side streams (default on).  Turn off when several
        handles already run concurrently on their own streams."""
        self._ck(self.lib.yn_multi_stream(self.h, int(bool(on))), "yn_multi_stream")

    def exact_f32(self, on=True):
        """Pin every GEMM-shaped conv to the f32 MFMA (default off: the MFMA-bound layers use split-f16 operands, fp32-class)."""
        self._ck(self.lib.yn_exact_f32(self.h, int(bool(on))), "yn_exact_f32")

    def range_status(self):
        """(weights_exceed_f16, activation_overflow) of the split-f16 range guard (yn_range_status): the first says the handle fell
        back to the f32-MFMA family at fold time; the second that an activation >= 65504 was split since the last call — the
        results of those calls are invalid, re-run them with exact_f32(True).  Synchronises the stream, clears the second flag."""
        w, a = _i32(0), _i32(0)
        self._ck(self.lib.yn_range_status(self.h, ctypes.byref(w), ctypes.byref(a)), "yn_range_status")
        return bool(w.value), bool(a.value)

    def fuse_decode(self, on=True):
        """infer(): last head conv + candidate decode as one kernel (default on; bit-identical outputs either way)."""
        self._ck(self.lib.yn_fuse_decode(self.h, int(on)), "yn_fuse_decode")      # 0 off, 1 when the heads are large enough, 2 always

    def nms_prefilter(self, mode=1):
        """First-chunk prefilter of the per-class NMS: 0 off, 1 for batches of >= 4 images (default), 2 always; same kept sets."""
        self._ck(self.lib.yn_nms_prefilter(self.h, int(mode)), "yn_nms_prefilter")

    def nms_sweep(self, on=True):
        """Spread-out large class segments on nms_sweep_kernel (x-extent broad phase, exact predicate) instead of the dense tiles (default on)."""
        self._ck(self.lib.yn_nms_sweep(self.h, int(bool(on))), "yn_nms_sweep")

    def nms_sweep_segments(self, B, C):
        """Testing aid: segments of the last NMS call (B images, C classes) that nms_sweep_kernel handled."""
        return int(self.lib.yn_nms_sweep_segments(self.h, int(B), int(C)))

    def down_fuse(self, on=True):
        """Main branch of the stride-2 unit of stage 2 as one kernel (default on; bit-identical outputs either way)."""
        self._ck(self.lib.yn_down_fuse(self.h, int(bool(on))), "yn_down_fuse")

    def tail_fuse(self, on=True):
        """Layers .2-.4 of the detection heads + the decode as one grouped kernel (default on; bit-identical outputs either way)."""
        self._ck(self.lib.yn_tail_fuse(self.h, int(bool(on))), "yn_tail_fuse")

    def group_launch(self, on=True):
        """The three heads' layer k / the three laterals as one grouped launch each (default on; bit-identical outputs either way)."""
        self._ck(self.lib.yn_group_launch(self.h, int(bool(on))), "yn_group_launch")

    def pw_config_count(self):
        return int(self.lib.yn_pw_config_count())

    def pw_f32_config_count(self):
        return int(self.lib.yn_pw_f32_config_count())

    def pw_families(self):
        """The two families of pointwise tile configurations: indices of the f32-MFMA one, indices of the split-f16 one."""
        n, nf = self.pw_config_count(), self.pw_f32_config_count()
        return list(range(nf)), list(range(nf, n))

    def synchronize(self):
        self._ck(self.lib.yn_synchronize(self.h), "yn_synchronize")

    # ---- weights
    def load_param(self, key, value):
        """value: numpy array / CPU tensor (host copy) or a GPU tensor (device copy)."""
        if isinstance(value, torch.Tensor) and value.is_cuda:
            v = value.detach()
            if v.dtype != torch.float32 and v.dtype != torch.int64:
                v = v.float()
            v = v.contiguous()
            shape = (ctypes.c_int64 * max(v.dim(), 1))(*v.shape)
            self._ck(self.lib.yn_load_param_dev(self.h, key.encode(), v.data_ptr(), shape, v.dim()), "yn_load_param_dev(%s)" % key)
            return
        import numpy as np
        a = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
        if a.dtype != np.int64:
            a = np.ascontiguousarray(a, dtype=np.float32)
        shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
        self._ck(self.lib.yn_load_param(self.h, key.encode(), a.ctypes.data, shape, a.ndim), "yn_load_param(%s)" % key)

    def load_state_dict(self, sd):
        for k, v in sd.items():
            self.load_param(k, v)

    def fold_bn(self):
        self._ck(self.lib.yn_fold_bn(self.h), "yn_fold_bn")

    def get_folded(self, conv_key, weight_shape):
        import numpy as np
        w = np.empty(weight_shape, dtype=np.float32)
        b = np.empty((weight_shape[0],), dtype=np.float32)
        self._ck(self.lib.yn_get_folded(self.h, conv_key.encode(), w.ctypes.data, b.ctypes.data), "yn_get_folded")
        return w, b

    # ---- network
    def head_shapes(self, B):
        return [(B, self.S // s, self.S // s, self.head_ch) for s in arch.STRIDES]

    def forward_raw(self, x, out=None):
        """x: cuda float32 NCHW [B,3,S,S] -> three NHWC head tensors."""
        B = x.shape[0]
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape[1:]) == (3, self.S, self.S), (x.shape, self.S)
        x = self._in(x)
        if out is None:
            out = [torch.empty(s, dtype=torch.float32, device=x.device) for s in self.head_shapes(B)]
        self._ck(self.lib.yn_forward_raw(self.h, x.data_ptr(), B, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()), "yn_forward_raw")
        return out

    def forward_taps(self, x):
        """The backbone taps (c3, c4, c5) of ShuffleNetV2.forward, NHWC."""
        B, S = x.shape[0], self.S
        ch = arch.STAGE_CH[self.backbone]
        outs = [torch.empty((B, S // st, S // st, c), dtype=torch.float32, device=x.device) for st, c in zip((8, 16, 32), ch)]
        self._ck(self.lib.yn_forward_taps(self.h, self._in(x).data_ptr(), B, *[o.data_ptr() for o in outs]), "yn_forward_taps")
        return outs

    def score_full(self, heads):
        B = heads[0].shape[0]
        N = self.N
        bbox = torch.empty((B, N, 4), dtype=torch.float32, device=heads[0].device)
        cls = torch.empty((B, N, self.C), dtype=torch.float32, device=heads[0].device)
        self._ck(self.lib.yn_score_full(self.h, _ptr(heads[0]), _ptr(heads[1]), _ptr(heads[2]), B, bbox.data_ptr(), cls.data_ptr()), "yn_score_full")
        return bbox, cls

    def decode_boxes(self, txtytwth):
        t = self._in(txtytwth)
        B = t.shape[0]
        out = torch.empty((B, self.N, 4), dtype=torch.float32, device=t.device)
        self._ck(self.lib.yn_decode_boxes(self.h, t.data_ptr(), B, out.data_ptr()), "yn_decode_boxes")
        return out

    def create_grid(self, S):
        import numpy as np
        HW = sum((S // s) ** 2 for s in arch.STRIDES)
        g = np.empty((1, HW, 1, 2), np.float32)
        st = np.empty((1, HW, self.A, 2), np.float32)
        an = np.empty((1, HW, self.A, 2), np.float32)
        self._ck(self.lib.yn_create_grid(self.h, int(S), g.ctypes.data, st.ctypes.data, an.ctypes.data), "yn_create_grid")
        return g, st, an

    # ---- post-processing
    def nms(self, dets, scores, thresh, diou=False):
        n = int(scores.shape[0])
        keep = torch.empty((max(n, 1),), dtype=torch.int32, device=dets.device)
        cnt = torch.zeros((1,), dtype=torch.int32, device=dets.device)
        self._ck(self.lib.yn_nms(self.h, _ptr(self._in(dets)), _ptr(self._in(scores)), n, float(thresh), int(bool(diou)),
                                 keep.data_ptr(), cnt.data_ptr()), "yn_nms")
        return keep[: int(cnt.item())]

    def alloc_outputs(self, B, N=None, device=None):
        N = self.N if N is None else N
        device = self.device if device is None else device
        return (torch.empty((B, N, 4), dtype=torch.float32, device=device),
                torch.empty((B, N), dtype=torch.float32, device=device),
                torch.empty((B, N), dtype=torch.int32, device=device),
                torch.empty((B, N), dtype=torch.int32, device=device),
                torch.zeros((B,), dtype=torch.int32, device=device))

    def postprocess(self, all_local, all_conf, out=None):
        """all_local [B,N,4], all_conf [B,N,C] cuda float32 -> (boxes, scores, cls, index, count) device buffers"""
        B, N, C = all_conf.shape
        out = self.alloc_outputs(B, N, all_conf.device) if out is None else out
        self._ck(self.lib.yn_postprocess(self.h, _ptr(self._in(all_local)), _ptr(self._in(all_conf)), B, N, C,
                                         *[o.data_ptr() for o in out]), "yn_postprocess")
        return out

    def infer(self, x, out=None):
        B = x.shape[0]
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape[1:]) == (3, self.S, self.S), (x.shape, self.S)
        x = self._in(x)
        out = self.alloc_outputs(B, device=x.device) if out is None else out
        self._ck(self.lib.yn_infer(self.h, x.data_ptr(), B, *[o.data_ptr() for o in out]), "yn_infer")
        return out                                              # out[4] (counts) NEGATIVE = range mark (yn_range_status): check before slicing with it

    def pack_detections(self, out, rec=None, offsets=None):
        """Kept rows of all images of `out` (infer / postprocess outputs) as one record list rec [B*N, 6] = x1,y1,x2,y2,score,class
        (first offsets[B] rows valid) + offsets [B+1] int32, both on the device (yn_pack_detections)."""
        boxes, scores, cls, _, count = out
        B, N = scores.shape
        rec = torch.empty((B * N, 6), dtype=torch.float32, device=scores.device) if rec is None else rec
        offsets = torch.empty((B + 1,), dtype=torch.int32, device=scores.device) if offsets is None else offsets
        self._ck(self.lib.yn_pack_detections(self.h, boxes.data_ptr(), scores.data_ptr(), cls.data_ptr(), count.data_ptr(), B, N,
                                             rec.data_ptr(), offsets.data_ptr()), "yn_pack_detections")
        return rec, offsets

    def detections_to_host(self, out):
        """models/yolo_nano.py:370-376 for a whole batch with two device-to-host copies: -> list of B (bboxes [K,4] f32,
        scores [K] f32, cls_inds [K] i64) numpy triples, fresh and writable."""
        try:
            rec, offsets = self.pack_detections(out)
            off = offsets.cpu().numpy()
            if int(off[-1]) < 0:                                # compact_kernel's range mark, carried through pack_kernel
                raise YnRangeError("yn_infer: an activation exceeded the split-f16 range (|x| >= 65504); results invalid, re-run under exact_f32")
        except YnRangeError:
            self.range_status()                                 # the exception IS the report: acknowledge, so that the flag does not poison later calls
            raise
        host = rec[: int(off[-1])].cpu().numpy()
        res = []
        for b in range(len(off) - 1):
            r = host[off[b]:off[b + 1]]
            res.append((r[:, :4].copy(), r[:, 4].copy(), r[:, 5].astype(np.int64)))
        return res

    # ---- training loss
    def loss(self, conf, cls, txtytwth, target, grads=True):
        """tools.loss + iou_score + decode on the reference's split prediction layout.
        -> (losses [4] device tensor, (g_conf, g_cls, g_txtytwth) or None)"""
        B = cls.shape[0]
        conf, cls, t, target = (self._in(v) for v in (conf, cls, txtytwth, target))
        losses = torch.empty((4,), dtype=torch.float32, device=cls.device)
        g = (torch.empty_like(conf), torch.empty_like(cls), torch.empty_like(t)) if grads else (None, None, None)
        self._ck(self.lib.yn_loss(self.h, conf.data_ptr(), cls.data_ptr(), t.data_ptr(), target.data_ptr(), B, losses.data_ptr(),
                                  _ptr(g[0]), _ptr(g[1]), _ptr(g[2])), "yn_loss")
        return losses, (g if grads else None)

    def loss_heads(self, heads, target, grads=True):
        """Same, directly on the three raw NHWC head tensors; gradients come back in the head layout."""
        B = heads[0].shape[0]
        target = self._in(target)
        losses = torch.empty((4,), dtype=torch.float32, device=target.device)
        g = [torch.empty_like(t) for t in heads] if grads else [None, None, None]
        self._ck(self.lib.yn_loss_heads(self.h, _ptr(heads[0]), _ptr(heads[1]), _ptr(heads[2]), target.data_ptr(), B, losses.data_ptr(),
                                        _ptr(g[0]), _ptr(g[1]), _ptr(g[2])), "yn_loss_heads")
        return losses, (g if grads else None)

    def sgd_step(self, params, grads, momentum_buf, lr, momentum=0.9, weight_decay=5e-4, grad_scale=1.0, first_step=False):
        """In-place SGD on flat float32 buffers (train.py:167-171); grad_scale = 1/world_size after a sum all-reduce."""
        assert params.is_contiguous() and grads.is_contiguous() and momentum_buf.is_contiguous() and params.numel() == grads.numel() == momentum_buf.numel()
        self._ck(self.lib.yn_sgd_step(self.h, params.data_ptr(), grads.data_ptr(), momentum_buf.data_ptr(), params.numel(),
                                      float(lr), float(momentum), float(weight_decay), float(grad_scale), int(bool(first_step))), "yn_sgd_step")

    def preprocess(self, img_u8, rw, rh, left, top, side, mean, std, out=None):
        """ValTransforms on the device: uint8 [h0,w0,3] BGR CUDA tensor -> float32 [3,side,side] (yn_preprocess)."""
        assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3 and img_u8.is_contiguous() and img_u8.is_cuda
        h0, w0 = int(img_u8.shape[0]), int(img_u8.shape[1])
        if out is None:
            out = torch.empty((3, side, side), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == 3 * side * side
        m = (ctypes.c_float * 3)(*[float(v) for v in mean])
        sd = (ctypes.c_float * 3)(*[float(v) for v in std])
        self._ck(self.lib.yn_preprocess(self.h, img_u8.data_ptr(), h0, w0, int(rw), int(rh), int(left), int(top), int(side),
                                        ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p), out.data_ptr()), "yn_preprocess")
        return out

    def preprocess_batch(self, imgs_u8, geoms, side, mean, std, out=None):
        """n images in one launch per 32 (yn_preprocess_batch): imgs_u8 = list of uint8 [h0,w0,3] CUDA tensors, geoms = list of
        (rw, rh, left, top) -> float32 [n,3,side,side]."""
        n = len(imgs_u8)
        if out is None:
            out = torch.empty((n, 3, side, side), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == n * 3 * side * side
        ptrs = (ctypes.c_void_p * max(n, 1))()
        geom = (ctypes.c_int32 * (6 * max(n, 1)))()
        for i, (im, g) in enumerate(zip(imgs_u8, geoms)):
            assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3 and im.is_contiguous() and im.is_cuda
            ptrs[i] = im.data_ptr()
            geom[6 * i:6 * i + 6] = [int(im.shape[0]), int(im.shape[1]), int(g[0]), int(g[1]), int(g[2]), int(g[3])]
        m = (ctypes.c_float * 3)(*[float(v) for v in mean])
        sd = (ctypes.c_float * 3)(*[float(v) for v in std])
        self._ck(self.lib.yn_preprocess_batch(self.h, n, ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(geom, ctypes.c_void_p), int(side),
                                              ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p), out.data_ptr()), "yn_preprocess_batch")
        return out

    def nms_merge(self, boxes, scores, cls, num_classes, nms_thresh, diou=False):
        """Per-class NMS over a detection list (TTA merge, utils/misc.py:132-146) -> (boxes [K,4], scores [K], cls [K], index [K])."""
        n = int(boxes.shape[0])
        boxes = self._in(boxes); scores = self._in(scores); cls = self._in(cls, torch.int32)
        ob = torch.empty((max(n, 1), 4), dtype=torch.float32, device=self.device)
        osc = torch.empty((max(n, 1),), dtype=torch.float32, device=self.device)
        oc = torch.empty((max(n, 1),), dtype=torch.int32, device=self.device)
        oi = torch.empty((max(n, 1),), dtype=torch.int32, device=self.device)
        cnt = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self._ck(self.lib.yn_nms_merge(self.h, boxes.data_ptr(), scores.data_ptr(), cls.data_ptr(), n, int(num_classes), float(nms_thresh),
                                       int(bool(diou)), ob.data_ptr(), osc.data_ptr(), oc.data_ptr(), oi.data_ptr(), cnt.data_ptr()), "yn_nms_merge")
        k = int(cnt.item())
        return ob[:k], osc[:k], oc[:k], oi[:k]

    def ema_update(self, ema, model, decay):
        """ema = ema * d + (1 - d) * model, in place (utils/misc.py:83-86); float32 tensors of equal size on this device."""
        assert ema.is_contiguous() and model.is_contiguous() and ema.numel() == model.numel() and ema.dtype == model.dtype == torch.float32
        self._ck(self.lib.yn_ema_update(self.h, ema.data_ptr(), model.data_ptr(), ema.numel(), float(decay)), "yn_ema_update")

    # ---- training labels
    def make_targets(self, label_lists, anchor_size, out=None):
        """tools.multi_gt_creator (tools.py:97-216): list (per image) of [xmin, ymin, xmax, ymax, class] rows ->
        float32 device tensor [B, N, 11].  The (tiny) label table crosses PCIe once; the assignment runs on the GPU."""
        import numpy as np
        B = len(label_lists)
        counts = [len(l) for l in label_lists]
        flat = np.zeros((max(sum(counts), 1), 5), dtype=np.float64)
        if sum(counts):
            flat[:sum(counts)] = np.array([row for l in label_lists for row in l], dtype=np.float64).reshape(-1, 5)
        offs = np.zeros(B + 1, dtype=np.int32)
        offs[1:] = np.cumsum(counts)
        anchors = np.ascontiguousarray(np.array(anchor_size, dtype=np.float64).reshape(-1))
        if anchors.size != 6 * self.A:
            raise YnError("anchor_size must hold %d [w,h] pairs" % (3 * self.A))
        d_flat = torch.from_numpy(flat).to(self.device, non_blocking=True)
        d_offs = torch.from_numpy(offs).to(self.device, non_blocking=True)
        if out is None:
            out = torch.empty((B, self.N, 11), dtype=torch.float32, device=self.device)
        self._ck(self.lib.yn_make_targets(self.h, d_flat.data_ptr(), d_offs.data_ptr(), B, anchors.ctypes.data, out.data_ptr()), "yn_make_targets")
        return out

    # ---- training step
    def train_bind(self):
        """Allocate the flat parameter / gradient / momentum buffers and bind them (parameters = the loaded state dict)."""
        n = int(self.lib.yn_train_param_count(self.h))
        self.flat_params = torch.empty(n, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.empty(n, dtype=torch.float32, device=self.device)
        self.flat_momentum = torch.empty(n, dtype=torch.float32, device=self.device)
        self._ck(self.lib.yn_train_bind(self.h, self.flat_params.data_ptr(), self.flat_grads.data_ptr(), self.flat_momentum.data_ptr(), n), "yn_train_bind")
        return n

    def train_precision(self, dtype="f32"):
        """Arithmetic of train_step: "f32" (the reference's own) or "f16" (fp16 storage + f16 MFMA, fp32 master weights, loss scaling)."""
        self._ck(self.lib.yn_train_precision(self.h, {"f32": 0, "fp32": 0, "f16": 1, "fp16": 1}[dtype]), "yn_train_precision")

    def allreduce_grads(self, nccl_comm):
        """All-reduce(sum) the flat gradient buffer in place over an RCCL communicator (an ncclComm_t as an int / c_void_p) on the
        handle's stream: the torch-free form of the gradient exchange (yn_allreduce_grads)."""
        self._ck(self.lib.yn_allreduce_grads(self.h, ctypes.c_void_p(nccl_comm if isinstance(nccl_comm, int) else nccl_comm.value)), "yn_allreduce_grads")

    def loss_scale(self):
        """(scale, clean steps) of the fp16 step's dynamic loss scale (yn_train_get_loss_scale; synchronises)."""
        s, c = _f32(0), _f32(0)
        self._ck(self.lib.yn_train_get_loss_scale(self.h, ctypes.byref(s), ctypes.byref(c)), "yn_train_get_loss_scale")
        return float(s.value), float(c.value)

    def set_loss_scale(self, scale, clean_steps=0.0):
        """Restore the loss scale (and its clean-step counter) of a checkpoint (yn_train_set_loss_scale)."""
        self._ck(self.lib.yn_train_set_loss_scale(self.h, float(scale), float(clean_steps)), "yn_train_set_loss_scale")

    def skipped_steps(self):
        n = ctypes.c_int64(0)
        self._ck(self.lib.yn_train_skipped_steps(self.h, ctypes.byref(n)), "yn_train_skipped_steps")
        return int(n.value)

    def head_fork(self, force=None):
        """The fp16 step's head-tower fork decision (yn_train_head_fork): None = query, False / True = pin it.  -> -1 undecided, 0, 1."""
        d = ctypes.c_int(-1)
        self._ck(self.lib.yn_train_head_fork(self.h, 0 if force is None else (2 if force else 1), ctypes.byref(d)), "yn_train_head_fork")
        return int(d.value)

    def param_slice(self, key):
        off, num = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self.lib.yn_train_param_offset(self.h, key.encode(), ctypes.byref(off), ctypes.byref(num)), "yn_train_param_offset")
        return slice(off.value, off.value + num.value)

    def train_step(self, x, target, lr=1e-3, momentum=0.9, weight_decay=5e-4, grad_scale=1.0, update=True):
        """-> losses [4] (conf, cls, bbox, iou) device tensor; gradients are left in self.flat_grads."""
        B = x.shape[0]
        x, target = self._in(x), self._in(target)
        losses = torch.empty((4,), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_train_step(self.h, x.data_ptr(), target.data_ptr(), B, float(lr), float(momentum), float(weight_decay),
                                        float(grad_scale), int(bool(update)), losses.data_ptr()), "yn_train_step")
        return losses

    def train_forward(self, x):
        """Train-mode forward only (batch statistics; running statistics updated) -> three NHWC float32 raw heads."""
        B = x.shape[0]
        outs = [torch.empty(s, dtype=torch.float32, device=x.device) for s in self.head_shapes(B)]
        self._ck(self.lib.yn_train_forward(self.h, self._in(x).data_ptr(), B, *[o.data_ptr() for o in outs]), "yn_train_forward")
        return outs

    def read_param(self, key, shape):
        import numpy as np
        a = np.empty(shape, dtype=np.float32)
        self._ck(self.lib.yn_read_param(self.h, key.encode(), a.ctypes.data, a.size), "yn_read_param(%s)" % key)
        return a

    # ---- measurement
    def profile_enable(self, on=True):
        self._ck(self.lib.yn_profile_enable(self.h, int(bool(on))), "yn_profile_enable")

    def profile_records(self):
        """[(layer name, kernel symbol, ms, algorithmic flops, algorithmic bytes)] of the last profiled call."""
        n = self.lib.yn_profile_count(self.h)
        recs = []
        name = ctypes.create_string_buffer(96)
        kern = ctypes.create_string_buffer(96)
        ms, fl, by = _f32(), ctypes.c_double(), ctypes.c_double()
        for i in range(n):
            self._ck(self.lib.yn_profile_get(self.h, i, name, 96, kern, 96, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)), "yn_profile_get")
            recs.append((name.value.decode(), kern.value.decode(), ms.value, fl.value, by.value))
        return recs

    # ---- single operators (NHWC device tensors; weights in torch layout on the device) -----------
    def op_dwconv3x3(self, x, w, bias, stride=1, act=0):
        B, H, W, C = x.shape
        y = torch.empty((B, (H - 1) // stride + 1, (W - 1) // stride + 1, C), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_op_dwconv3x3(self.h, _ptr(self._in(x)), B, H, W, C, stride, _ptr(self._in(w)),
                                          _ptr(self._in(bias)) if bias is not None else None, act, y.data_ptr()), "yn_op_dwconv3x3")
        return y

    def op_pwconv(self, x, w, bias, act=0):
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_op_pwconv(self.h, _ptr(self._in(x)), B, H, W, Cin, Cout, _ptr(self._in(w)),
                                       _ptr(self._in(bias)) if bias is not None else None, act, y.data_ptr()), "yn_op_pwconv")
        return y

    def op_pwconv_shuffle(self, x, passthrough, w, bias, act=0):
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        y = torch.empty((B, H, W, 2 * Cout), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_op_pwconv_shuffle(self.h, self._in(x).data_ptr(), self._in(passthrough).data_ptr(), B, H, W, Cin, Cout,
                                               self._in(w).data_ptr(), _ptr(bias), act, y.data_ptr()), "yn_op_pwconv_shuffle")
        return y

    def op_conv3x3(self, x, w, bias, act=0, x2=None, resample=0):
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_op_conv3x3(self.h, _ptr(self._in(x)), _ptr(self._in(x2)) if x2 is not None else None, resample,
                                        B, H, W, Cin, Cout, _ptr(self._in(w)),
                                        _ptr(self._in(bias)) if bias is not None else None, act, y.data_ptr()), "yn_op_conv3x3")
        return y

    def op_stem(self, x_nchw, w, bias, act=0):
        B, _, H, W = x_nchw.shape
        Cout = w.shape[0]
        y = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout), dtype=torch.float32, device=x_nchw.device)
        self._ck(self.lib.yn_op_stem(self.h, _ptr(self._in(x_nchw)), B, H, W, Cout, _ptr(self._in(w)),
                                     _ptr(self._in(bias)) if bias is not None else None, act, y.data_ptr()), "yn_op_stem")
        return y

    def op_maxpool(self, x):
        B, H, W, C = x.shape
        y = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_op_maxpool3x3s2(self.h, _ptr(self._in(x)), B, H, W, C, y.data_ptr()), "yn_op_maxpool3x3s2")
        return y

    def op_shuffle_block(self, block, x, cout, stride):
        B, H, W, _ = x.shape
        y = torch.empty((B, (H - 1) // stride + 1, (W - 1) // stride + 1, cout), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_op_shuffle_block(self.h, block.encode(), _ptr(self._in(x)), B, H, W, y.data_ptr()), "yn_op_shuffle_block")
        return y

    def op_h16_conv(self, kind, x, w, bias=None, stride=1, dy=None, gapped=False):
        """One conv kernel of the fp16 training step (+ its two gradient kernels when dy is given): x [B,H,W,Cin] fp32 NHWC,
        kind 0 pw / 1 dw / 2 dense3x3 -> (y, dx, dw) fp32 (dx, dw None without dy)."""
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if dy is not None else None
        dw = torch.empty_like(w) if dy is not None else None
        self._ck(self.lib.yn_op_h16_conv(self.h, int(kind), self._in(x).data_ptr(), B, H, W, Cin, int(bool(gapped)), self._in(w).data_ptr(), _ptr(bias),
                                         Cout, int(stride), _ptr(self._in(dy) if dy is not None else None), y.data_ptr(), _ptr(dx), _ptr(dw)), "yn_op_h16_conv")
        return y, dx, dw

    def op_h16_bn(self, y, gamma, beta, act=0, dz=None):
        M, C = y.shape
        z = torch.empty_like(y)
        dy = torch.empty_like(y) if dz is not None else None
        dg = torch.empty((C,), dtype=torch.float32, device=y.device) if dz is not None else None
        db = torch.empty((C,), dtype=torch.float32, device=y.device) if dz is not None else None
        self._ck(self.lib.yn_op_h16_bn(self.h, self._in(y).data_ptr(), _ptr(self._in(dz) if dz is not None else None), M, C, gamma.data_ptr(), beta.data_ptr(),
                                       int(act), z.data_ptr(), _ptr(dy), _ptr(dg), _ptr(db)), "yn_op_h16_bn")
        return z, dy, dg, db

    def train_graph(self, enable=None):
        """Switch the fp16 step's hipGraph replay (None: leave); -> number of steps served from a graph so far."""
        n = ctypes.c_int64(0)
        self._ck(self.lib.yn_train_graph(self.h, -1 if enable is None else int(bool(enable)), ctypes.byref(n)), "yn_train_graph")
        return int(n.value)

    def op_h16_gemm_stats(self, kind, x, w, gapped=False, dy=None, y_below=None, mean=None, invstd=None, gamma=None, beta=None, act=0):
        """hgemm with its HColStat epilogue: -> y, sums_fwd (numpy double [2][Cout]) and, with dy, dx, sums_bwd ([2][Cin])."""
        import numpy as np
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
        sf = np.zeros((2, Cout), np.float64)
        dx = torch.empty_like(x) if dy is not None else None
        sb = np.zeros((2, Cin), np.float64) if dy is not None else None
        keep = [self._in(t) if t is not None else None for t in (x, w, dy, y_below, mean, invstd, gamma, beta)]
        self._ck(self.lib.yn_op_h16_gemm_stats(self.h, int(kind), keep[0].data_ptr(), B, H, W, Cin, int(bool(gapped)), keep[1].data_ptr(), Cout, y.data_ptr(),
                                               sf.ctypes.data, _ptr(keep[2]), _ptr(keep[3]), _ptr(keep[4]), _ptr(keep[5]), _ptr(keep[6]), _ptr(keep[7]), int(act),
                                               _ptr(dx), sb.ctypes.data if sb is not None else None), "yn_op_h16_gemm_stats")
        return y, sf, dx, sb

    def op_h16_bn_unit(self, y, passthrough, gamma, beta, act=0, dunit=None):
        """BatchNorm as the last layer of a ShuffleV2 unit: -> unit [M][2C] (and dy, deven, dgamma, dbeta when dunit [M][2C] is given)."""
        M, C = y.shape
        unit = torch.empty((M, 2 * C), dtype=torch.float32, device=y.device)
        mk = lambda *shape: torch.empty(shape, dtype=torch.float32, device=y.device) if dunit is not None else None
        dy, dev, dg, db = mk(M, C), mk(M, C), mk(C), mk(C)
        yc, pc, dc = self._in(y), self._in(passthrough), (self._in(dunit) if dunit is not None else None)
        self._ck(self.lib.yn_op_h16_bn_unit(self.h, yc.data_ptr(), pc.data_ptr(), _ptr(dc), M, C, gamma.data_ptr(), beta.data_ptr(), int(act),
                                            unit.data_ptr(), _ptr(dy), _ptr(dev), _ptr(dg), _ptr(db)), "yn_op_h16_bn_unit")
        return unit, dy, dev, dg, db

    def to_nhwc(self, x):
        B, C, H, W = x.shape
        y = torch.empty((B, H, W, C), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_op_nchw_to_nhwc(self.h, _ptr(self._in(x)), B, C, H, W, y.data_ptr()), "yn_op_nchw_to_nhwc")
        return y

    def to_nchw(self, x):
        B, H, W, C = x.shape
        y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        self._ck(self.lib.yn_op_nhwc_to_nchw(self.h, _ptr(self._in(x)), B, C, H, W, y.data_ptr()), "yn_op_nhwc_to_nchw")
        return y
