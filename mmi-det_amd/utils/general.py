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
    [xyxy, conf, cls] per image, by decreasing confidence), computed by mmi_nms on the device; the only host round trip in
    the common case is reading the B detection counts to slice the result.

    labels (general.py:519-526, the autolabelling path of test.py --save-hybrid): per image a (n,5) tensor [cls, x, y, w, h];
    each becomes a candidate row with objectness 1 and class score 1, appended behind the image's predictions.
    max_nms (general.py:501,557-559): when an image has more than 30 000 candidates -- (row, class) pairs under multi_label,
    rows otherwise -- only the 30 000 most confident take part.  That needs the 30 000th-best confidence, so it costs one extra
    host round trip, taken only when the candidate count CAN exceed the cap (rows x classes > 30 000: test.py's
    conf_thres = 0.001 with multi_label on a 640x640 image does)."""
    from mmidet_hip import ops
    max_nms = 30000
    pred = prediction.float()
    nc = pred.shape[2] - 5
    if labels and any(len(l) for l in labels):
        lmax = max(len(l) for l in labels)
        extra = torch.zeros((pred.shape[0], lmax, nc + 5), dtype=pred.dtype, device=pred.device)
        for xi, l in enumerate(labels):
            if len(l):
                l = l.to(pred.device, pred.dtype)
                extra[xi, :len(l), :4] = l[:, 1:5]
                extra[xi, :len(l), 4] = 1.0
                extra[xi, torch.arange(len(l), device=pred.device), l[:, 0].long() + 5] = 1.0
        pred = torch.cat((pred, extra), 1)
    ml = bool(multi_label) and nc > 1
    if pred.shape[1] * (nc if ml else 1) > max_nms:
        live = pred[..., 4:5] > conf_thres
        score = pred[..., 5:] * pred[..., 4:5]                        # conf = obj_conf * cls_conf
        cand = ((score > conf_thres) & live) if ml else ((score.max(2, keepdim=True).values > conf_thres) & live)
        if classes is not None:
            allow = torch.zeros(nc, dtype=torch.bool, device=pred.device)
            allow[[int(c) for c in classes if 0 <= int(c) < nc]] = True
            cand = cand & (allow.view(1, 1, -1) if ml else allow[score.argmax(2, keepdim=True)])
        over = cand.flatten(1).sum(1) > max_nms                       # (the extra host round trip)
        if bool(over.any()):
            pred = pred.clone()
            flat = torch.where(cand, score if ml else score.max(2, keepdim=True).values, score.new_zeros(())).flatten(1)
            kth = flat.topk(max_nms, dim=1).values[:, -1]             # the 30 000th-best confidence per image
            for xi in torch.nonzero(over).flatten().tolist():
                if ml:
                    drop = cand[xi] & (score[xi] < kth[xi])
                    pred[xi, :, 5:][drop] = 0.0                       # the pair no longer clears conf_thres
                else:
                    drop = cand[xi, :, 0] & (score[xi].max(1).values < kth[xi])
                    pred[xi, drop, 4] = 0.0
    out, nout = ops.nms(pred, float(conf_thres), float(iou_thres), classes, bool(agnostic), bool(multi_label))
    counts = nout.tolist()
    return [out[i, :n] for i, n in enumerate(counts)]


# names this module does not define (the reference's helpers outside the hot path) come from the reference checkout's
# module of the same name when one is overlaid: mmidet_hip/overlay.py
from mmidet_hip.overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__)
