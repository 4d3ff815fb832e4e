"""Small host-side helpers of the hot path (reference: utils/general.py make_divisible 230-232, bbox_iou 403-447)."""
import math

import torch


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def bbox_iou(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """API-compatible IoU helper for callers outside the training step (tensor ops on whatever device the boxes live).
    The training loss does NOT go through this function: ComputeLoss runs the fused HIP loss kernel."""
    box2 = box2.T
    if x1y1x2y2:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1[0], box1[1], box1[2], box1[3]
        b2_x1, b2_y1, b2_x2, b2_y2 = box2[0], box2[1], box2[2], box2[3]
    else:
        b1_x1, b1_x2, b1_y1, b1_y2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2, box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
        b2_x1, b2_x2, b2_y1, b2_y2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2, box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    iw = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0)
    ih = (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    inter = iw * ih
    w1, h1, w2, h2 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps, b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if not (GIoU or DIoU or CIoU):
        return iou
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    if GIoU:
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    if DIoU:
        return iou - rho2 / c2
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def xywh2xyxy(x):
    """utils/general.py:392-400 of the reference."""
    y = x.clone()
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=()):
    """Reference utils/general.py:486-580 with the same signature and return type (list of (n,6) tensors
    [xyxy, conf, cls] per image, by decreasing confidence), computed by mmi_nms on the device; the only host round trip
    is reading the B detection counts to slice the result."""
    from mmidet_hip import ops
    if labels:
        raise NotImplementedError('a-priori labels (autolabelling, general.py:519-526) are outside the detection path')
    out, nout = ops.nms(prediction.float(), float(conf_thres), float(iou_thres), classes, bool(agnostic), bool(multi_label))
    counts = nout.tolist()
    return [out[i, :n] for i, n in enumerate(counts)]


# names this module does not define (the reference's helpers outside the hot path) come from the reference checkout's
# module of the same name when one is overlaid: mmidet_hip/overlay.py
from mmidet_hip.overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__)
