"""yolo-nano_amd — MI355X-native YOLO-Nano hot path (imported as ``yolo_nano_amd``)."""
from . import arch, weights  # noqa: F401
