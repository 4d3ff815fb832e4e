"""Diagnostic: per-parameter gradient error of the HIP path vs an fp64 run of the oracle (tiny FFM graph)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, 'mmi-det_amd'), os.path.join(REPO, 'tests')]
from conftest import tiny_cfg  # noqa: E402
from oracle import portable_init  # noqa: E402
from oracle.ref_loss import ComputeLoss as OLoss, scaled_hyp  # noqa: E402
from oracle.ref_model import Model as OModel  # noqa: E402
from models.yolo_test import Model  # noqa: E402
from utils.loss import ComputeLoss  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'fourier'
cfg = tiny_cfg(kind)
nc = cfg['nc']
imgs, tg = portable_init.synth_batch(2, 128, nc, per_image=4, seed=int(os.environ.get("SEED", 1)))


def oracle(dtype):
    o = OModel(cfg, dropout=0.0)
    o.load_state_dict(portable_init.fill_(o.state_dict()))
    o = o.to(dtype)
    o.nc, o.gr, o.hyp = nc, 1.0, scaled_hyp(nc, 128)
    x = (imgs.float() / 255).to(dtype)
    o.train()
    pred, comb = o(x[:, :3], x[:, 3:])
    l, _ = OLoss(o)(pred, tg.to(dtype), comb.reshape(-1))
    l.backward()
    return {n: p.grad.double() for n, p in o.named_parameters() if p.grad is not None}, o.state_dict()


g64, sd = oracle(torch.float64)
g32, _ = oracle(torch.float32)
m = Model(tiny_cfg(kind))
m.load_state_dict({k: v.float() for k, v in portable_init.fill_(OModel(cfg).state_dict()).items()}, strict=True)
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
m = m.cuda().train()
m.nc, m.gr, m.hyp = nc, 1.0, scaled_hyp(nc, 128)
x = imgs.cuda().float() / 255
pred, comb = m(x[:, :3], x[:, 3:])
l, _ = ComputeLoss(m)(pred, tg.cuda(), comb.reshape(-1))
l.backward()
rows = []
for n, p in m.named_parameters():
    if p.grad is None or float(g64[n].norm()) < 1e-5:
        continue
    ref = g64[n]
    e_gpu = float((p.grad.double().cpu() - ref).norm() / ref.norm())
    e_cpu = float((g32[n] - ref).norm() / ref.norm())
    rows.append((e_gpu, e_cpu, n, float(ref.norm())))
rows.sort(reverse=True)
print('median gpu %.2e cpu %.2e' % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
for r in rows[:25]:
    print('%.2e (cpu %.2e) |g|=%.2e %s' % (r[0], r[1], r[3], r[2]))
kinds = {}
for e, c, n, _ in rows:
    k = 'gpt' if 'trans_blocks' in n or 'ln_f' in n or 'pos_emb' in n else ('bn' if '.bn' in n else 'conv')
    kinds.setdefault(k, []).append((e, c))
for k, v in kinds.items():
    print(k, len(v), 'median gpu %.2e cpu %.2e' % (np.median([a for a, _ in v]), np.median([b for _, b in v])))

# ---- per-layer output-gradient comparison (where does the backward error first appear?) -------------------------
def layer_grads(model, run, to_nchw):
    store = {}
    hooks = []

    def mk(i):
        def fwd_hook(mod, inp, out):
            if torch.is_tensor(out) and out.requires_grad:
                out.register_hook(lambda g, i=i: store.__setitem__(i, to_nchw(g.detach())))
        return fwd_hook
    for i, mod in enumerate(model.model):
        hooks.append(mod.register_forward_hook(mk(i)))
    run()
    for h in hooks:
        h.remove()
    return store


o = OModel(cfg, dropout=0.0)
o.load_state_dict(portable_init.fill_(o.state_dict()))
o = o.double()
o.nc, o.gr, o.hyp = nc, 1.0, scaled_hyp(nc, 128)
o.train()
xd = (imgs.float() / 255).double()


def run_o():
    pred, comb = o(xd[:, :3], xd[:, 3:])
    OLoss(o)(pred, tg.double(), comb.reshape(-1))[0].backward()


def run_m():
    m.zero_grad()
    pred, comb = m(x[:, :3], x[:, 3:])
    ComputeLoss(m)(pred, tg.cuda(), comb.reshape(-1))[0].backward()


so = layer_grads(o, run_o, lambda g: g)
sm = layer_grads(m, run_m, lambda g: g.permute(0, 3, 1, 2).double().cpu() if g.dim() == 4 else g.double().cpu())
print('layer output-gradient errors (HIP fp32 vs oracle fp64):')
for i in sorted(so.keys(), reverse=True):
    if i in sm and so[i].shape == sm[i].shape:
        print('  layer %2d %-28s %.2e' % (i, type(o.model[i]).__name__, float((sm[i] - so[i]).norm() / so[i].norm())))
