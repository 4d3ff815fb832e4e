"""Overlay package: the native hot-path modules (yolo_test, yolo, common, experimental) first; every other `models.*`
module of an MMI-Det checkout resolves to the reference's own file (mmidet_hip/overlay.py)."""
from mmidet_hip import overlay as _overlay

_overlay.extend_package_path(__name__, __path__)
