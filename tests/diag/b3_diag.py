"""Per-parameter gradient error of the tiny fourier graph in split-bf16 mode vs fp32 mode, both against the oracle."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, 'mmi-det_amd'), os.path.join(R, 'tests')]
from mmidet_hip import lib  # noqa: E402
from oracle import portable_init  # noqa: E402
from oracle.ref_loss import ComputeLoss as OLoss  # noqa: E402
from test_model_gpu import build_pair  # noqa: E402
from test_ops_gpu import dev, rel_err  # noqa: E402
from utils.loss import ComputeLoss  # noqa: E402

for seed in (2, 3, 4):
    for mode in (0, 1, 2):
        lib.set_gemm_precision(mode)
        m, o, cfg = build_pair('fourier', 128)
        imgs, targets = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=seed)
        x = imgs.float() / 255
        m.train(); o.train()
        po, co = o(x[:, :3], x[:, 3:])
        lo, io = OLoss(o)(po, targets, co.reshape(-1)); lo.backward()
        xd = x.to(dev())
        pg, cg = m(xd[:, :3], xd[:, 3:])
        lg, ig = ComputeLoss(m)(pg, targets.to(dev()), cg.reshape(-1)); lg.backward()
        og = dict(o.named_parameters())
        errs = sorted(((rel_err(p.grad, og[n].grad), n) for n, p in m.named_parameters()
                       if og[n].grad is not None and p.grad is not None and float(og[n].grad.norm()) > 1e-5), reverse=True)
        med = errs[len(errs) // 2][0]
        print('seed %d mode %d: loss err %.1e  worst %.2e %s | 2nd %.2e %s | median %.1e | preds %.1e' % (
            seed, mode, rel_err(lg, lo), errs[0][0], errs[0][1], errs[1][0], errs[1][1], med, rel_err(pg[0], po[0])), flush=True)
lib.set_gemm_precision(0)
