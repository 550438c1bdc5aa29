"""yolo-nano_amd — MI355X-native YOLO-Nano hot path (imported as ``yolo_nano_amd``).

``arch`` / ``weights`` are torch-free; ``YOLONano`` and ``fuse_conv_bn`` (the drop-in surface of
models/yolo_nano.py and utils/fuse_conv_bn.py) are imported lazily so that the pure-python parts
stay usable without torch.
"""
from . import arch, weights  # noqa: F401


def __getattr__(name):
    if name in ("YOLONano", "fuse_conv_bn", "Conv", "ShuffleNetV2", "ShuffleV2Block", "shufflenetv2", "SGD", "multi_gt_creator", "ModelEMA", "TestTimeAugmentation", "ValTransforms", "rescale_boxes"):
        from . import model
        return getattr(model, name)
    if name in ("Handle", "YnError", "YnRangeError", "load_library"):
        from . import capi
        return getattr(capi, name)
    raise AttributeError(name)
