"""BASELINE.json names ``models.yolo.Model``; the reference's two-stream class actually lives in models/yolo_test.py
(train.py:27 imports it from there).  Both import paths resolve to the same MI355X-native implementation."""
from models.yolo_test import Detect, Model, parse_model  # noqa: F401
