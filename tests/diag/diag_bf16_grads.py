"""Diagnostic (GPU): per-layer gradient error of the bf16-storage mode against the fp32 mode on the tiny graphs, with the
BatchNorm layers in training mode and frozen.  Noise accumulated through the depth grows from the head towards the stem;
a kernel bug would not.    python tests/diag/diag_bf16_grads.py [add|fourier]"""
import copy
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..', '..', 'mmi-det_amd'))
from test_model_gpu import build_pair  # noqa: E402
from test_ops_gpu import rel_err  # noqa: E402


def run(kind, frozen, bs, size):
    from oracle import portable_init
    from utils.loss import ComputeLoss
    m32, _, cfg = build_pair(kind, size)
    mbf = copy.deepcopy(m32)
    mbf.storage = 'bf16'
    imgs, targets = portable_init.synth_batch(bs, size, cfg['nc'], per_image=4, seed=11)
    x = (imgs.float() / 255).cuda()
    for m in (m32, mbf):
        m.train()
        if frozen:
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.eval()
        p, c = m(x[:, :3], x[:, 3:])
        loss, _ = ComputeLoss(m)(p, targets.cuda(), c.reshape(-1))
        loss.backward()
    torch.cuda.synchronize()
    print('== %s  BN %s  B=%d %dx%d' % (kind, 'frozen' if frozen else 'train', bs, size, size))
    by_layer = {}
    for (n, p), q in zip(m32.named_parameters(), mbf.parameters()):
        if p.grad is None or float(p.grad.norm()) < 1e-9:
            continue
        e = rel_err(q.grad, p.grad)
        cos = float((q.grad.double() * p.grad.double()).sum() / (q.grad.double().norm() * p.grad.double().norm() + 1e-30))
        layer = n.split('.')[1] if n.startswith('model.') else n.split('.')[0]
        by_layer.setdefault(layer, []).append((e, cos))
    for layer in sorted(by_layer, key=lambda k: (0, int(k)) if k.isdigit() else (1, k)):
        v = by_layer[layer]
        es = sorted(a for a, _ in v)
        cs = sorted(b for _, b in v)
        print('layer %-8s tensors %3d  rel err median %.3f max %.3f   cosine median %.3f min %.3f' %
              (layer, len(v), es[len(es) // 2], es[-1], cs[len(cs) // 2], cs[0]))


if __name__ == '__main__':
    kind = sys.argv[1] if len(sys.argv) > 1 else 'add'
    run(kind, False, 2, 128)
    run(kind, True, 2, 128)
    run(kind, False, 8, 256)
