"""GPU parity over EVERY graph the reference ships (models/transformer/*.yaml of the reference, 18 files; one carries the reference's
own unparsable typo row): the native model against the CPU oracle on the same hash weights and synthetic batch -- training forward,
loss and its four items, the auxiliary values, every parameter gradient, BatchNorm running statistics -- at reduced width and depth
(the graph TOPOLOGY is what differs between the files: where the streams are fused, Add against the fusion transformers, one or
three transformer stages, the FFM stage and its place, 1280-style extra levels), so that each file's lane / twin / concat / fan-out
plan runs once under the parity gate.  The two graphs of tests/golden (fuse3_fourier, fusion_add_vedai) are pinned against the
reference itself in test_model_gpu.py; the oracle they pin is the one used here."""
import copy
import glob
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import CFG_DIR
from test_ops_gpu import close, dev, rel_err

pytestmark = pytest.mark.gpu

FILES = sorted(os.path.basename(f) for f in glob.glob(os.path.join(CFG_DIR, '*.yaml'))
               if os.path.basename(f) != 'yolov5l_fusion_transformer_FLIR_aligned.yaml')   # (the reference's typo row: unparsable there too)


def tiny(name):
    with open(os.path.join(CFG_DIR, name)) as f:
        d = yaml.safe_load(f)
    d['depth_multiple'], d['width_multiple'] = 0.33, 0.25
    for row in d['backbone']:
        if row[2] == 'GPT1_fourier':
            row[3] = [32]                      # reference quirk B3: the FFM width is not scaled by width_multiple
    return d


@pytest.mark.parametrize('name', FILES)
def test_graph_matches_the_oracle(name):
    from models.yolo_test import Model
    from oracle import portable_init
    from oracle.ref_loss import ComputeLoss as OLoss, scaled_hyp
    from oracle.ref_model import Model as OModel
    from utils.loss import ComputeLoss
    cfg = tiny(name)
    nc = cfg['nc']
    o = OModel(copy.deepcopy(cfg), dropout=0.0)
    sd = portable_init.fill_(o.state_dict())
    o.load_state_dict(sd)
    m = Model(copy.deepcopy(cfg))
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for mm in (o, m):
        mm.nc, mm.gr, mm.hyp = nc, 1.0, scaled_hyp(nc, 128)
    m = m.to(dev()).train()
    o.train()
    imgs, targets = portable_init.synth_batch(2, 128, nc, per_image=4, seed=2)
    x = imgs.float() / 255
    po, co = o(x[:, :3], x[:, 3:])
    lo, io = OLoss(o)(po, targets, co.reshape(-1))
    lo.backward()
    xd = x.to(dev())
    pg, cg = m(xd[:, :3], xd[:, 3:])
    lg, ig = ComputeLoss(m)(pg, targets.to(dev()), cg.reshape(-1))
    lg.backward()
    torch.cuda.synchronize()
    assert len(pg) == len(po)
    for i in range(len(po)):
        close(pg[i], po[i], what='pred%d' % i)
    assert tuple(lg.shape) == tuple(lo.shape)
    close(lg, lo, what='loss', tol=1e-4)
    close(ig, io, what='loss items', tol=1e-4)
    assert cg.numel() == co.numel()
    if co.numel():
        close(cg, co, what='Combine_loss', tol=1e-4)
        for attr, tol in (('SSIMloss', 1e-4), ('PTLoss', 1e-3), ('Entropy_loss', 5e-3), ('ContrastiveValue', 1e-5)):
            a, b = getattr(m, attr), getattr(o, attr)
            if torch.is_tensor(b) and b.numel():
                close(a, b, what=attr, tol=tol)
    og = dict(o.named_parameters())
    names = [n for n, p in m.named_parameters() if p.grad is not None]
    assert names == [n for n, p in o.named_parameters() if p.grad is not None]          # the same parameters are trained
    errs = [(rel_err(p.grad, og[n].grad), n) for n, p in m.named_parameters()
            if p.grad is not None and 'key_proj.bias' not in n and float(og[n].grad.norm()) > 1e-5]
    assert len(errs) > 50
    worst = max(errs)
    assert worst[0] < 5e-3, worst          # (the bound of test_every_parameter_gradient_vs_oracle)
    assert np.median([e for e, _ in errs]) < 5e-4
    osd, msd = o.state_dict(), m.state_dict()
    for k, v in osd.items():
        if k.endswith('running_mean') or k.endswith('running_var'):
            close(msd[k], v, what=k, tol=1e-4)
    # evaluation mode on the updated running statistics: the decoded detections and the raw head outputs (test.py:123-139), and
    # again after Model.fuse() (BatchNorm folded into the convolutions, the lane form instead of twin launches)
    m.eval()
    o.eval()
    with torch.no_grad():
        (zo, po_e), _ = o(x[:, :3], x[:, 3:])
        (zg, pg_e), _ = m(xd[:, :3], xd[:, 3:])
        close(zg, zo, what='eval: decoded detections')
        for i in range(len(po_e)):
            close(pg_e[i], po_e[i], what='eval: head output %d' % i)
        m.fuse()
        (zf, _), _ = m(xd[:, :3], xd[:, 3:])
        close(zf, zo, what='eval after fuse(): decoded detections', tol=2e-3)
