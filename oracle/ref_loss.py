"""Oracle (test infrastructure, CPU/fp32 torch): restatement of utils/loss.py ComputeLoss and
utils/general.py bbox_iou(CIoU) of the reference.  Prints of the reference (loss.py:162,172,173,182) are dropped.
"""
import math

import torch
import torch.nn as nn

# data/hyp.scratch.yaml:6-22 (only the keys the loss reads)
HYP_SCRATCH = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0)


def scaled_hyp(nc, imgsz=640, nl=3, hyp=None):
    """train.py:689-691: box*=3/nl; cls*=nc/80*3/nl; obj*=(imgsz/640)^2*3/nl."""
    h = dict(HYP_SCRATCH if hyp is None else hyp)
    h['box'] *= 3. / nl
    h['cls'] *= nc / 80. * 3. / nl
    h['obj'] *= (imgsz / 640) ** 2 * 3. / nl
    return h


def bbox_ciou(box1, box2, eps=1e-7):
    """CIoU of xywh boxes; box1 is (4,n), box2 is (n,4).  utils/general.py:403-447 with x1y1x2y2=False, CIoU=True."""
    box2 = box2.T
    b1_x1, b1_x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
    b1_y1, b1_y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
    b2_x1, b2_x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
    b2_y1, b2_y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * \
            (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


class ComputeLoss:
    """utils/loss.py:87-245.  ``model`` needs .hyp, .gr and .model[-1] = Detect."""

    def __init__(self, model, autobalance=False):
        device = next(model.parameters()).device
        h = model.hyp
        self.BCEcls = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([h['cls_pw']], device=device))
        self.BCEobj = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([h['obj_pw']], device=device))
        self.cp, self.cn = 1.0 - 0.5 * h.get('label_smoothing', 0.0), 0.5 * h.get('label_smoothing', 0.0)
        assert h['fl_gamma'] == 0, 'FocalLoss is outside the path (hyp.scratch fl_gamma=0)'
        det = model.model[-1]
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, .02])
        self.gr, self.hyp = model.gr, h
        self.na, self.nc, self.nl, self.anchors = det.na, det.nc, det.nl, det.anchors

    def __call__(self, p, targets, CombineLoss, alpha_Contrast=0.1, Flag=True):  # loss.py:113-184
        device = targets.device
        lcls, lbox, lobj = (torch.zeros(1, device=device) for _ in range(3))
        tcls, tbox, indices, anchors = self.build_targets(p, targets)
        for i, pi in enumerate(p):
            b, a, gj, gi = indices[i]
            tobj = torch.zeros_like(pi[..., 0], device=device)
            n = b.shape[0]
            if n:
                ps = pi[b, a, gj, gi]
                pxy = ps[:, :2].sigmoid() * 2. - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anchors[i]
                iou = bbox_ciou(torch.cat((pxy, pwh), 1).T, tbox[i])
                lbox += (1.0 - iou).mean()
                tobj[b, a, gj, gi] = (1.0 - self.gr) + self.gr * iou.detach().clamp(0).type(tobj.dtype)
                if self.nc > 1:
                    t = torch.full_like(ps[:, 5:], self.cn, device=device)
                    t[range(n), tcls[i]] = self.cp
                    lcls += self.BCEcls(ps[:, 5:], t)
            lobj += self.BCEobj(pi[..., 4], tobj) * self.balance[i]
        lbox *= self.hyp['box']
        lobj *= self.hyp['obj']
        lcls *= self.hyp['cls']
        bs = tobj.shape[0]
        det = lbox + lobj + lcls
        if Flag:
            if len(CombineLoss) == 0:                                   # :163-164
                avg = torch.zeros(1, device=device)
            else:                                                       # :167
                avg = sum(CombineLoss) / len(CombineLoss) * alpha_Contrast
            loss = torch.unsqueeze(avg, dim=0) + det                    # :171,175 -> (1,1) when CombineLoss is empty
        else:
            loss = det
        return loss * bs, torch.cat((lbox, lobj, lcls, det)).detach()

    def build_targets(self, p, targets):  # loss.py:189-245
        na, nt = self.na, targets.shape[0]
        tcls, tbox, indices, anch = [], [], [], []
        gain = torch.ones(7, device=targets.device)
        ai = torch.arange(na, device=targets.device).float().view(na, 1).repeat(1, nt)
        targets = torch.cat((targets.repeat(na, 1, 1), ai[:, :, None]), 2)
        g = 0.5
        off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], device=targets.device).float() * g
        for i in range(self.nl):
            anchors, shape = self.anchors[i], p[i].shape
            gain[2:6] = torch.tensor(p[i].shape)[[3, 2, 3, 2]]
            t = targets * gain
            if nt:
                r = t[:, :, 4:6] / anchors[:, None]
                j = torch.max(r, 1. / r).max(2)[0] < self.hyp['anchor_t']
                t = t[j]
                gxy = t[:, 2:4]
                gxi = gain[[2, 3]] - gxy
                j, k = ((gxy % 1. < g) & (gxy > 1.)).T
                l, m = ((gxi % 1. < g) & (gxi > 1.)).T
                j = torch.stack((torch.ones_like(j), j, k, l, m))
                t = t.repeat((5, 1, 1))[j]
                offsets = (torch.zeros_like(gxy)[None] + off[:, None])[j]
            else:
                t = targets[0]
                offsets = 0
            b, c = t[:, :2].long().T
            gxy = t[:, 2:4]
            gwh = t[:, 4:6]
            gij = (gxy - offsets).long()
            gi, gj = gij.T
            a = t[:, 6].long()
            indices.append((b, a, gj.clamp_(0, shape[2] - 1), gi.clamp_(0, shape[3] - 1)))
            tbox.append(torch.cat((gxy - gij, gwh), 1))
            anch.append(anchors[a])
            tcls.append(c)
        return tcls, tbox, indices, anch
