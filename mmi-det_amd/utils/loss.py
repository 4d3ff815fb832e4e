"""ComputeLoss on the MI355X-native kernels: drop-in for the reference's utils/loss.py ComputeLoss (87-245) as called at
train.py:707,789 and test.py:135.

__call__(p, targets, CombineLoss, alpha_Contrast=0.1, Flag=True) -> (loss * bs, cat(lbox, lobj, lcls, Detectloss).detach())
build_targets(p, targets) -> (tcls, tbox, indices, anch)

Target assignment is one integer HIP kernel (bit-exact, no host sync inside __call__); matching/CIoU/BCE, their
gradients and all gains are one fused pass (csrc/loss.hip).  The reference's prints (loss.py:162,172,173,182) are not
reproduced (each is a device sync).  FocalLoss (fl_gamma > 0) and autobalance are outside the path.
"""
import ctypes

import torch
from torch.autograd import Function

from mmidet_hip import alloc, lib, loss_ops
from mmidet_hip.ops import _stream, scratch
from utils.torch_utils import is_parallel


def smooth_BCE(eps=0.1):
    return 1.0 - 0.5 * eps, 0.5 * eps


def _is_head_view(p):
    """(B,na,ny,nx,no) as the permuted view of a dense NHWC (B,ny,nx,na*no) tensor (Detect.forward in training mode)."""
    if p.dim() != 5:
        return False
    b, na, ny, nx, no = p.shape
    return p.stride() == (ny * nx * na * no, no, nx * na * no, na * no, 1)


class _DetectLoss(Function):
    @staticmethod
    def forward(ctx, cfg, targets, combine, *preds):
        (anchors, grids, na, nc, balance, hbox, hobj, hcls, gr, cp, cn, anchor_t, alpha, flag) = cfg
        nl = len(preds)
        # Detect hands over its (B,na,ny,nx,no) tensors as strided views of the head convolutions' NHWC outputs (no permute copy):
        # the kernels read and write that layout directly; anything else is made contiguous, the layout of the reference
        nhwc = all(_is_head_view(p) for p in preds)
        if not nhwc:
            preds = [p.contiguous() for p in preds]
        dev = preds[0].device
        bs = preds[0].shape[0]
        idx, tcls, tbox, anch, counts, cap = loss_ops.build_targets_raw(targets, anchors, grids, anchor_t)
        dps = [alloc.empty_like(p) for p in preds]
        total = sum(p.numel() // p.shape[-1] for p in preds)
        nbytes = lib.detect_loss_workspace(nl, total, cap)
        ws = scratch(nbytes // 4 + 4, dev, slot=3)
        out5 = alloc.empty(5, dtype=torch.float32, device=dev)
        pp = (ctypes.c_void_p * nl)(*[p.data_ptr() for p in preds])
        dpp = (ctypes.c_void_p * nl)(*[d.data_ptr() for d in dps])
        gh = (ctypes.c_int32 * (2 * nl))(*[v for g in grids for v in g])
        bh = (ctypes.c_float * nl)(*balance)
        ncomb = 0 if combine is None else combine.numel()
        lib.detect_loss(pp, dpp, gh, nl, bs, na, nc, idx.data_ptr(), tcls.data_ptr(), tbox.data_ptr(), anch.data_ptr(),
                        counts.data_ptr(), cap, bh, hbox, hobj, hcls, gr, cp, cn,
                        combine.data_ptr() if ncomb else None, ncomb, alpha, (1 if flag else 0) | (2 if nhwc else 0), ws.data_ptr(), nbytes,
                        out5.data_ptr(), _stream())
        ctx.save_for_backward(*dps)
        ctx.mark_non_differentiable(out5)
        return out5.narrow(0, 0, 1).clone(), out5

    @staticmethod
    def backward(ctx, g, _g5):
        outs = []
        g = g.contiguous()
        for dp in ctx.saved_tensors:
            o = alloc.empty_like(dp)
            lib.scale(dp.data_ptr(), g.data_ptr(), o.data_ptr(), dp.numel(), _stream())
            outs.append(o)
        return (None, None, None, *outs)


class ComputeLoss:
    def __init__(self, model, autobalance=False):
        assert not autobalance, 'autobalance is outside the hot path'
        h = model.hyp
        assert h.get('fl_gamma', 0.0) == 0, 'FocalLoss (fl_gamma > 0) is outside the hot path'
        assert h.get('cls_pw', 1.0) == 1.0 and h.get('obj_pw', 1.0) == 1.0, 'pos_weight != 1 is outside the hot path'
        self.cp, self.cn = smooth_BCE(eps=h.get('label_smoothing', 0.0))
        det = model.module.model[-1] if is_parallel(model) else model.model[-1]
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, .02])
        self.ssi = 0
        self.gr, self.hyp, self.autobalance = model.gr, h, autobalance
        for k in 'na', 'nc', 'nl', 'anchors':
            setattr(self, k, getattr(det, k))

    def _grids(self, p):
        return [(int(pi.shape[2]), int(pi.shape[3])) for pi in p]

    def __call__(self, p, targets, CombineLoss, alpha_Contrast=0.1, Flag=True):
        self.CombineLoss = CombineLoss
        targets = targets.to(p[0].device, torch.float32)
        comb = None
        if CombineLoss is not None and len(CombineLoss) > 0:       # len() of a 0-d tensor raises, as in the reference
            comb = CombineLoss.detach().to(p[0].device, torch.float32).contiguous()
        cfg = (self.anchors.to(p[0].device), self._grids(p), self.na, self.nc, self.balance[:self.nl],
               float(self.hyp['box']), float(self.hyp['obj']), float(self.hyp['cls']), float(self.gr), float(self.cp),
               float(self.cn), float(self.hyp['anchor_t']), float(alpha_Contrast), bool(Flag))
        loss, out5 = _DetectLoss.apply(cfg, targets, comb, *p)
        if Flag and comb is None:
            loss = loss.unsqueeze(0)                                # (1,1): reference quirk when CombineLoss is empty
        return loss, out5[1:5].detach()

    def build_targets(self, p, targets):
        targets = targets.to(p[0].device, torch.float32)
        return loss_ops.build_targets(targets, self.anchors.to(p[0].device), self._grids(p), self.hyp['anchor_t'])


# names this module does not define (the reference's helpers outside the hot path) come from the reference checkout's
# module of the same name when one is overlaid: mmidet_hip/overlay.py
from mmidet_hip.overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__)
