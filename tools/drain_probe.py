"""How long do the wgrad side streams keep running after the lanes have finished their backward?  Events on the lane stream and
on every wgrad stream at the moment backward has been enqueued, before the join: drain = latest side event - lane event."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'mmi-det_amd')]
import bench  # noqa: E402
from mmidet_hip import ops  # noqa: E402
from mmidet_hip.train_step import TrainStep  # noqa: E402
from models.yolo_test import Model  # noqa: E402

cfg = bench.load_cfg('l_fourier')
dev = torch.device('cuda:0')
model = Model(cfg).to(dev).train()
ts = TrainStep(model, cfg['nc'], 640, 16, accumulate=1)
imgs, tg = bench.synth(16, 640, cfg['nc'], dev, 1)
recs = []
orig = ops.join_pending


def probed():
    main = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(main)
    sides = []
    for sd in ops.side_streams_in_flight() + [model._ir_stream(dev)]:
        e = torch.cuda.Event(enable_timing=True)
        e.record(sd)
        sides.append(e)
    recs.append((e0, sides))
    orig()


ops.join_pending = probed
import mmidet_hip.train_step as T  # noqa: E402
for i in range(8):
    ts.step(imgs, tg)
torch.cuda.synchronize()
for e0, sides in recs[3:]:
    d = [e0.elapsed_time(e) for e in sides]
    print('after the lane stream is done: wgrad streams finish at %s ms, IR lane at %+.2f ms' % (', '.join('%+.2f' % v for v in d[:-1]), d[-1]))
