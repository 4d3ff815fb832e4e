"""Runtime helpers the hot path depends on (reference: utils/torch_utils.py initialize_weights 144-153, is_parallel,
model_info 204-227, ModelEMA 269-299)."""
import logging
import math
from copy import deepcopy

import torch
import torch.nn as nn

logger = logging.getLogger(__name__)


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel) or hasattr(model, 'module')


def intersect_dicts(da, db, exclude=()):
    """Keys of da that db has with the same shape, minus `exclude` substrings (reference utils/torch_utils.py:139-141;
    train.py:528 uses it to load a checkpoint into a re-configured model)."""
    return {k: v for k, v in da.items() if k in db and not any(x in k for x in exclude) and v.shape == db[k].shape}


def initialize_weights(model):
    """Parity-critical constants: every BatchNorm2d gets eps=1e-3, momentum=0.03."""
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif type(m) in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6):
            m.inplace = True


def fuse_conv_and_bn(conv, bn):
    """Inference folding of BatchNorm(eval) into the convolution (reference utils/torch_utils.py:181-201):
    w' = w * gamma / sqrt(var + eps) per output channel, b' = beta - mean * gamma / sqrt(var + eps) (+ the scaled conv
    bias if there was one).  The folded weight keeps the kernels' channels_last (OHWI) memory format."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.eps + bn.running_var)
        w = conv.weight.detach() * scale.view(-1, 1, 1, 1)
        fused.weight.data = w.contiguous(memory_format=torch.channels_last)
        b_conv = torch.zeros_like(scale) if conv.bias is None else conv.bias.detach()
        fused.bias.copy_(b_conv * scale + bn.bias - bn.running_mean * scale)
    return fused


def model_info(model, verbose=False, img_size=640):
    n_p = sum(x.numel() for x in model.parameters())
    n_g = sum(x.numel() for x in model.parameters() if x.requires_grad)
    logger.info("Model Summary: %d layers, %d parameters, %d gradients", len(list(model.modules())), n_p, n_g)


class ModelEMA:
    """Exponential moving average of every floating state_dict entry, decay ramp 0.9999*(1-exp(-n/2000))."""

    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(model.module if is_parallel(model) else model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        # deepcopy clones every parameter on its own, so a packed model's copy is unpacked again.  Pack the copy HERE, while
        # nobody has cached its addresses yet (FusedSGDEMA's pointer table, _pairs below); a forward of the copy never re-seats
        # anything (models/yolo_test.py::_pack_for_twin).
        if hasattr(self.ema, 'pack_parameters') and any(p.is_cuda for p in self.ema.parameters()):
            self.ema.pack_parameters()
        self._pairs = None

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            if self._pairs is None:
                msd = (model.module if is_parallel(model) else model).state_dict()
                self._pairs = [(v, msd[k]) for k, v in self.ema.state_dict().items() if v.dtype.is_floating_point]
            ema_t = [a for a, _ in self._pairs]
            torch._foreach_mul_(ema_t, d)
            torch._foreach_add_(ema_t, [b.detach() for _, b in self._pairs], alpha=1. - d)

    def update_attr(self, model, include=(), exclude=('process_group', 'reducer')):
        for k, v in model.__dict__.items():
            if (len(include) and k not in include) or k.startswith('_') or k in exclude:
                continue
            setattr(self.ema, k, v)


# names this module does not define (the reference's helpers outside the hot path) come from the reference checkout's
# module of the same name when one is overlaid: mmidet_hip/overlay.py
from mmidet_hip.overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__)
