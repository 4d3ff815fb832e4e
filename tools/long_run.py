"""Loss trajectory of the bench workload over many steps (same model/init/batch as bench.py): does training stay finite?"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'mmi-det_amd')]
import bench  # noqa: E402
from mmidet_hip import lib  # noqa: E402
from mmidet_hip.train_step import TrainStep  # noqa: E402
from models.yolo_test import Model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lib.set_gemm_precision(mode)
torch.manual_seed(2)
cfg = bench.load_cfg('l_fourier')
dev = torch.device('cuda:0')
model = Model(cfg).to(dev)
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.1
model.train()
ts = TrainStep(model, cfg['nc'], 640, bs, accumulate=1)
imgs, tg = bench.synth(bs, 640, cfg['nc'], dev, 100)
switch = {int(a.split(':')[0]): int(a.split(':')[1]) for a in sys.argv[4:]}      # e.g. 110:2 117:1 -> change mode at those steps
for it in range(steps):
    if it in switch:
        torch.cuda.synchronize()
        lib.set_gemm_precision(switch[it])
        print('-- gemm precision', switch[it], flush=True)
    loss, items = ts.step(imgs, tg)
    if it % 10 == 0 or it == steps - 1 or it in switch or (it - 1) in switch or not torch.isfinite(loss).all():
        l = items.tolist()
        gmax = 0.0
        print('step %3d  loss/bs %.4f  box %.4f obj %.4f cls %.4f  ssim %.4f' % (it, float(loss) / bs, l[0], l[1], l[2],
                                                                               float(model.SSIMloss)), flush=True)
        if not torch.isfinite(loss).all():
            bad = [k for k, v in model.state_dict().items() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
            print('non-finite state entries:', len(bad), bad[:8])
            break
