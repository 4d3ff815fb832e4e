"""SPP pools (5/9/13) forward and backward at the step's size (16 x 20 x 20 x 512, both lanes): ms per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
from mmidet_hip import ops  # noqa: E402

d = torch.device('cuda:0')
x = torch.randn(16, 20, 20, 512, device=d, requires_grad=True)
y = ops.spp_pool(x)
g = torch.randn_like(y)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def fb():
    x.grad = None
    ops.spp_pool(x).backward(g)


with torch.no_grad():
    tf = timed(lambda: ops.spp_pool(x))
tfb = timed(fb)
print('SPP forward %.1f us, forward + backward %.1f us (backward ~%.1f us)' % (tf * 1e3, tfb * 1e3, (tfb - tf) * 1e3))
