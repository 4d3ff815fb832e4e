"""Contour Enhancement Module at the step's size (B x 640 x 640 x 3): forward and forward+backward, fused forward kernel vs the
four-kernel chain (MMIDET_CEM_FUSED semantics, toggled in-process).  Prints ms per call and the kernels' algorithmic HBM bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
from mmidet_hip import ops  # noqa: E402
from models.common import AdaptiveModule3  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 640
d = torch.device('cuda:0')
torch.manual_seed(0)
m = AdaptiveModule3(3, 3).to(d).train()
x = torch.rand(B, H, W, 3, device=d)
gy = torch.randn(B, H, W, 3, device=d)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def fwd():
    with torch.no_grad():
        return m(x)


def fwd_bwd():
    for p in m.parameters():
        p.grad = None
    y = m(x)
    y.backward(gy)
    ops.join_pending()


def fwd_train():
    y = m(x)
    ops.join_pending()
    return y


px = B * H * W
rows = ((False, False, False), (True, False, False), (True, True, False), (True, True, True))
if os.environ.get('BENCH_CEM_FUSED_ONLY') == '1':
    rows = ((True, True, ops.CEM_BWD_BN),)          # (the configuration the environment selects; default: the shipped one)
for fused, bwd, bn in rows:
    ops.CEM_FUSED, ops.CEM_BWD_FUSED, ops.CEM_BWD_BN = fused, bwd, bn
    m.train()
    t_tr = timed(fwd_bwd)
    t_tf = timed(fwd_train)
    m.eval()
    t_ev = timed(fwd)
    m.train()
    print('fused forward=%d  fused backward middle=%d  + BN2 reduction=%d  train fwd+bwd %.3f ms   train fwd %.3f ms   eval fwd %.3f ms' % (fused, bwd, bn, t_tr, t_tf, t_ev))
# algorithmic bytes of the fused training forward: x twice (pre-pass + main) + y2, t (24 ch) + chansum + y3 written, then BN3+act: y3, x read, out written
fw = px * 4 * (3 + 3 + 24 + 24 + 1 + 3 + 3 + 3 + 3)
print('fused training forward, algorithmic HBM bytes: %.1f MB (%.3f ms at 8 TB/s)' % (fw / 1e6, fw / 8e12 * 1e3))
