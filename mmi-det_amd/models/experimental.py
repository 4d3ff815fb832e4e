"""Checkpoint loading for the MI355X-native model: drop-in for the reference's
models/experimental.py attempt_load (113-134) and Ensemble (97-110), as used by test.py:60 and detect_twostream.py:33.

The reference stores whole pickled module objects (train.py:881-899: `{'model': deepcopy(model).half(), 'ema': ...}`).
Unpickling resolves `models.yolo_test.Model`, `models.common.Conv` ... to THIS package's classes (same names), but the
unpickled object carries the attribute set of whoever wrote it.  So the weights are what is taken from it: a fresh native
Model is built from the stored `.yaml` and filled through `state_dict()` (identical keys and shapes), which
works for checkpoints written by the reference and by this package alike."""
import torch
import torch.nn as nn


class Ensemble(nn.ModuleList):
    """NMS ensemble of two-stream models: predictions concatenated along the box axis (reference experimental.py:97-110)."""

    def forward(self, x, x2, augment=False):
        y = [module(x, x2, augment)[0][0] for module in self]       # eval-mode det = (z, list): take z
        return torch.cat(y, 1), None


def model_from_checkpoint_object(obj, device=None):
    """A pickled Model object of either origin -> a native fp32 Model with the same weights, names and hyper-parameters."""
    from models.yolo_test import Model
    fresh = Model(obj.yaml)
    sd = {k: v.float() for k, v in obj.state_dict().items()}
    missing, unexpected = fresh.load_state_dict(sd, strict=False)
    unexpected = [k for k in unexpected if 'total_ops' not in k and 'total_params' not in k]   # thop leftovers (reference B5)
    if missing or unexpected:
        raise RuntimeError('checkpoint does not fit the graph it names: missing %s unexpected %s' % (missing[:5], unexpected[:5]))
    for k in ('names', 'nc', 'hyp', 'gr', 'class_weights'):
        if hasattr(obj, k):
            setattr(fresh, k, getattr(obj, k))
    return fresh.to(device) if device is not None else fresh


def attempt_load(weights, map_location=None, fuse=True):
    """weights = path or [paths] of reference-format checkpoints -> fp32, Conv+BN-fused, eval-mode model (or Ensemble), as
    the reference's `ckpt[...].float().fuse().eval()`; fuse=False keeps the BatchNorm layers (e.g. to go on training)."""
    model = Ensemble()
    for w in weights if isinstance(weights, (list, tuple)) else [weights]:
        ckpt = torch.load(w, map_location='cpu', weights_only=False)
        obj = ckpt['ema' if ckpt.get('ema') else 'model'] if isinstance(ckpt, dict) else ckpt
        m = model_from_checkpoint_object(obj, device=map_location)
        m = m.float()
        model.append((m.fuse() if fuse else m).eval())
    if len(model) == 1:
        return model[-1]
    for k in ('names', 'stride'):
        setattr(model, k, getattr(model[-1], k))
    return model


# names this module does not define (the reference's helpers outside the hot path) come from the reference checkout's
# module of the same name when one is overlaid: mmidet_hip/overlay.py
from mmidet_hip.overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__)
