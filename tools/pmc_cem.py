"""Two training calls of the Contour Enhancement Module at the step's size: the target of `rocprofv3 --pmc` passes (tools/pmc_cem.sh)."""
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
for _ in range(2):
    for p in m.parameters():
        p.grad = None
    y = m(x)
    y.backward(gy)
    ops.join_pending()
torch.cuda.synchronize()
