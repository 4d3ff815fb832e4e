"""fp64 arbiter for the full-size gradients (test infrastructure; build container only: imports the REAL reference).

    python oracle/gen_fp64_fullsize.py          # -> tests/golden/fullsize_fp64.npz  (~100 KB)

Runs joewybean/MMI-Det's own Model + ComputeLoss in float64 on the BASELINE graph at its real layer shapes (yolov5l
two-stream-fourier, nc=6, 640x640, batch 2, hash-initialised weights, dropout 0, the synthetic batch of
tests/test_model_gpu.py::test_yolov5l_640_train_step_matches_oracle) and stores, per parameter tensor, the L2 norm of
its fp64 gradient and two fixed +-1 projections of it (oracle/portable_init.py::signs), plus the loss and checksums of the
three head outputs.  The GPU test rebuilds the fp64 gradients with the oracle (which this file pins at full size: the two
must agree to fp64 rounding) and then asks, for EVERY parameter tensor, whether the HIP gradient is as close to the fp64
truth as the CPU fp32 evaluation of the same graph is.  Only data is written."""
import contextlib
import io
import os
import sys
import time
from copy import deepcopy

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import portable_init  # noqa: E402
from oracle.gen_golden import REF, import_reference  # noqa: E402
from oracle.ref_loss import scaled_hyp  # noqa: E402

NPROJ = 2


def full_cfg():
    import yaml
    with open(os.path.join(REF, 'models/transformer/yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml')) as f:
        cfg = yaml.safe_load(f)
    cfg['nc'] = 6
    return cfg


def projections(g, name):
    """NPROJ dot products of a gradient tensor with hash-generated +-1 vectors (float64)."""
    flat = g.detach().double().reshape(-1)
    out = []
    for k in range(NPROJ):
        s = portable_init.signs(flat.numel(), '%s#%d' % (name, k))
        out.append(float((flat * s).sum()))
    return out


def main():
    t0 = time.time()
    Model, ComputeLoss = import_reference()[:2]
    torch.set_default_dtype(torch.float64)
    with contextlib.redirect_stdout(io.StringIO()):
        model = Model(deepcopy(full_cfg())).double()   # (a few of the reference's parameters are created as float32 explicitly)
    sd = model.state_dict()
    portable_init.fill_(sd)                      # fp32 hash values, widened exactly
    model.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.nc, model.gr, model.hyp = 6, 1.0, scaled_hyp(6, 640)
    model.train()
    imgs, targets = portable_init.synth_batch(2, 640, 6, per_image=8, seed=3)
    x = (imgs.float() / 255).double()            # the fp32 input values, widened exactly
    with contextlib.redirect_stdout(io.StringIO()):
        pred, comb = model(x[:, :3], x[:, 3:])
        loss, items = ComputeLoss(model)(pred, targets.double(), comb.reshape(-1))
        loss.backward()
    names, norms, projs = [], [], []
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        norms.append(float(p.grad.norm()))
        projs.append(projections(p.grad, n))
    out = dict(names=np.array(names), norms=np.array(norms), projs=np.array(projs), loss=loss.detach().numpy(),
               items=items.detach().numpy(), combine=comb.detach().numpy(),
               pred_norms=np.array([float(p.norm()) for p in pred]), pred_sums=np.array([float(p.sum()) for p in pred]))
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'fullsize_fp64.npz'), **out)
    print('wrote tests/golden/fullsize_fp64.npz: %d tensors, loss %.12f, %.0f s' % (len(names), float(loss), time.time() - t0))


if __name__ == '__main__':
    main()
