"""Forward time per top-level layer (single stream, HIP events around each module of model.model), and the share of the
token (GPT / GPT1_fourier) blocks: where the sequential small-kernel chains sit."""
import os
import sys

os.environ['MMIDET_TWO_STREAMS'] = '0'
import torch  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'mmi-det_amd')]
import bench  # noqa: E402
from models.yolo_test import Model  # noqa: E402

cfg = bench.load_cfg('l_fourier')
dev = torch.device('cuda:0')
model = Model(cfg).to(dev).train()
imgs, tg = bench.synth(16, 640, cfg['nc'], dev, 1)
x = imgs.float() / 255
rgb, ir = x[:, :3], x[:, 3:]
recs = []


def pre(m, inp):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    m._t0 = e


def post(m, inp, out):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    recs.append((m.i, m.type.split('.')[-1], m._t0, e))


for m in model.model:
    m.register_forward_pre_hook(pre)
    m.register_forward_hook(post)
for it in range(3):
    recs.clear()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    with torch.no_grad():
        model(rgb, ir)
    t1.record()
    torch.cuda.synchronize()
tot = t0.elapsed_time(t1)
by = {}
for i, ty, a, b in recs:
    ms = a.elapsed_time(b)
    by[ty] = by.get(ty, 0.0) + ms
    print('%3d %-14s %7.3f ms' % (i, ty, ms))
print('forward total %.2f ms (single stream, no_grad)' % tot)
for ty, ms in sorted(by.items(), key=lambda kv: -kv[1]):
    print('  %-14s %7.2f ms  %4.1f%%' % (ty, ms, 100 * ms / tot))
