"""Which ATen kernels still run inside one training step, and from where: torch.profiler with Python stacks, every aten:: op
that launches a device kernel grouped by its innermost repo frames (engine-internal work -- gradient fan-out accumulation,
undefined-gradient materialisation -- shows up with the autograd node that triggered it)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'mmi-det_amd')]
import bench  # noqa: E402
from mmidet_hip.train_step import TrainStep  # noqa: E402
from models.yolo_test import Model  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'l_fourier'
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = bench.load_cfg(wl)
dev = torch.device('cuda:0')
model = Model(cfg).to(dev).train()
ts = TrainStep(model, cfg['nc'], 640, bs, accumulate=1)
imgs, tg = bench.synth(bs, 640, cfg['nc'], dev, 1)
for _ in range(2):
    ts.step(imgs, tg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(imgs, tg)
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith('aten::') or e.device_time_total <= 0:
        continue
    if any(c.name.startswith('aten::') and c.device_time_total > 0 for c in (e.cpu_children or [])):
        continue            # (count the innermost op that owns the kernel)
    frames = [f for f in (e.stack or []) if 'mmi-det_amd' in f or 'bench.py' in f][:3]
    seq = ''
    par = e.cpu_parent
    while par is not None:
        if 'Backward' in par.name or 'AccumulateGrad' in par.name or par.name.startswith('autograd::engine'):
            seq = par.name
            break
        par = par.cpu_parent
    k = (e.name, seq, ' <- '.join(f.split('mmi-det_amd/')[-1] for f in frames))
    by[k][0] += 1
    by[k][1] += e.device_time_total
print('%5s %9s  %s' % ('calls', 'device us', 'op | autograd node | python frames'))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print('%5d %9.1f  %s | %s | %s' % (v[0], v[1], *k))
