"""yolov5l 640x640 B=2: parameter-gradient error against the CPU oracle for the GEMM arithmetic modes (and fp32 with the
stream-K schedule off), to tell arithmetic error from the discrete events (max-pool ties) that any perturbation can trigger."""
import os
import sys

import torch
import yaml

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, 'mmi-det_amd'), os.path.join(R, 'tests')]
from mmidet_hip import lib  # noqa: E402
from models.yolo_test import Model  # noqa: E402
from oracle import portable_init  # noqa: E402
from oracle.ref_loss import ComputeLoss as OLoss, scaled_hyp  # noqa: E402
from oracle.ref_model import Model as OModel  # noqa: E402
from test_ops_gpu import dev, rel_err  # noqa: E402
from utils.loss import ComputeLoss  # noqa: E402

with open(os.path.join(R, 'mmi-det_amd', 'models', 'transformer', 'yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml')) as f:
    cfg = yaml.safe_load(f)
cfg['nc'] = 6
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
o = OModel(cfg, dropout=0.0)
sd = portable_init.fill_(o.state_dict())
o.load_state_dict(sd)
o.nc, o.gr, o.hyp = 6, 1.0, scaled_hyp(6, 640)
o.train()
imgs, targets = portable_init.synth_batch(2, 640, 6, per_image=8, seed=seed)
x = imgs.float() / 255
po, co = o(x[:, :3], x[:, 3:])
lo, io = OLoss(o)(po, targets, co.reshape(-1))
lo.backward()
og = {n: p.grad for n, p in o.named_parameters()}
NAMES = ['Enhance.conv3.weight', 'Enhance.conv2.weight', 'model.1.conv.weight', 'model.4.conv.weight', 'model.10.m.4.cv2.conv.weight',
         'model.17.m.8.cv1.conv.weight', 'model.25.cv3.conv.weight', 'model.29.trans_blocks.3.mlp.0.weight', 'model.49.m.1.weight']
for label, mode, sk in (('fp32 MFMA', 0, 0), ('fp32 MFMA, stream-K off', 0, -1), ('bf16x6', 2, 0), ('bf16x9', 3, 0), ('bf16x3', 1, 0)):
    m = Model(cfg)
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.nc, m.gr, m.hyp = 6, 1.0, scaled_hyp(6, 640)
    m = m.to(dev()).train()
    lib.set_gemm_precision(mode)
    lib.set_streamk_slots(sk)
    xd = x.to(dev())
    pg, cg = m(xd[:, :3], xd[:, 3:])
    lg, ig = ComputeLoss(m)(pg, targets.to(dev()), cg.reshape(-1))
    lg.backward()
    torch.cuda.synchronize()
    lib.set_gemm_precision(0)
    lib.set_streamk_slots(0)
    g = dict(m.named_parameters())
    errs = sorted(rel_err(p.grad, og[n]) for n, p in g.items() if og[n] is not None and p.grad is not None and float(og[n].norm()) > 1e-7)
    print('%-26s loss %.1e pred %.1e | median %.1e p90 %.1e worst %.1e | ' % (label, rel_err(lg, lo), rel_err(pg[0], po[0]),
          errs[len(errs) // 2], errs[int(len(errs) * 0.9)], errs[-1]) + ' '.join('%.0e' % rel_err(g[n].grad, og[n]) for n in NAMES), flush=True)
    del m
print('columns:', ' '.join(n.replace('.weight', '') for n in NAMES))
