"""torch.profiler view of one training step (which ATen ops still run next to the HIP kernels, and how often)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    ts.step(imgs, tg)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=70))
print(prof.key_averages().table(sort_by='count', row_limit=30, max_name_column_width=70))
