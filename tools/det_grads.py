"""Run-to-run determinism of forward + backward: the same model, the same batch, R repetitions; every gradient tensor compared
bit for bit with the first repetition.  Over the launch-structure switches (two backbone lanes, wgrad side streams), so that a
difference points at a cross-stream ordering problem rather than at arithmetic.   python tools/det_grads.py [tiny|l] [size] [reps]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'mmi-det_amd'), os.path.join(REPO, 'tests')]
import bench  # noqa: E402
from mmidet_hip import ops  # noqa: E402
from mmidet_hip.train_step import HYP_SCRATCH, scale_hyp  # noqa: E402
from models.yolo_test import Model  # noqa: E402
from utils.loss import ComputeLoss  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
if kind == 'tiny':
    from conftest import tiny_cfg
    cfg = tiny_cfg('fourier')
    cfg['nc'] = 6
else:
    cfg = bench.load_cfg('l_fourier')
dev = torch.device('cuda:0')
imgs, tg = bench.synth(2, size, cfg['nc'], dev, 7)
x = imgs.float() / 255


def grads_of(model):
    for p in model.parameters():
        p.grad = None
    pred, comb = model(x[:, :3], x[:, 3:])
    loss, _ = ComputeLoss(model)(pred, tg, comb.reshape(-1))
    loss.backward()
    ops.join_pending()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


for lanes in (True, False):
    for overlap in (True, False):
        torch.manual_seed(0)
        model = Model(cfg).to(dev)
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        model.nc, model.gr, model.hyp = cfg['nc'], 1.0, scale_hyp(HYP_SCRATCH, cfg['nc'], size)
        model.train()
        model.two_streams = lanes
        ops.OVERLAP_WGRAD = overlap
        for m in model.modules():                       # identical BatchNorm state at every repetition
            if isinstance(m, torch.nn.BatchNorm2d):
                m.momentum = 0.0
        l0, g0 = grads_of(model)
        bad = {}
        for r in range(1, reps):
            l, g = grads_of(model)
            for n in g0:
                if not torch.equal(g0[n], g[n]):
                    bad.setdefault(n, []).append(float((g0[n] - g[n]).abs().max() / (g0[n].abs().max() + 1e-30)))
        names = list(g0)
        first = min((names.index(n) for n in bad), default=-1)
        print('lanes=%d overlap_wgrad=%d: loss %.9g, %d of %d gradient tensors differ between repetitions%s' % (
            lanes, overlap, l0, len(bad), len(g0), '' if not bad else '; first in module order: %s; worst relative difference %.2e' % (
                names[first], max(max(v) for v in bad.values()))), flush=True)
        if bad:
            for n in list(bad)[:12]:
                print('     ', n, ['%.1e' % v for v in bad[n]])
