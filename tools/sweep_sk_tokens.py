"""Stream-K forced on the long-K / few-tiles token GEMMs (MLP2 forward: 4C -> C, MLP1 dgrad: C <- 4C at M = 2048 rows): planner's
choice (one workgroup per 64x64 tile) against mmi_set_streamk_slots(n) for a few n."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mmidet_hip import lib, ops  # noqa: E402
from bench_conv import timeit  # noqa: E402

SHAPES = [(2048, 1, 1, 2048, 512, 1), (2048, 1, 1, 1024, 256, 1), (2048, 1, 1, 512, 128, 1), (2048, 1, 1, 4096, 1024, 1),
          (2048, 1, 1, 512, 2048, 1), (2048, 1, 1, 256, 1024, 1), (2048, 1, 1, 128, 512, 1), (2048, 1, 1, 1024, 4096, 1),
          (2048, 1, 1, 1024, 1024, 1), (2048, 1, 1, 512, 512, 1)]
SLOTS = [0, 256, 512, 768, 1024]
d = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
print('%-30s %s   (fwd ms / dgrad ms)' % ('shape', '   '.join('%15s' % ('planner' if v == 0 else 'SK %d' % v) for v in SLOTS)))
for (B, H, W, Ci, Co, k) in SHAPES:
    x = torch.randn(B, H, W, Ci, device=d)
    w = torch.randn(Co, k, k, Ci, device=d) * 0.05
    desc = ops._desc((B, H, W, Ci), Co, k, 1, Ci, Co)
    y = torch.empty(B, desc.Ho, desc.Wo, Co, device=d)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    cells = []
    ref = None
    for v in SLOTS:
        lib.set_streamk_slots(v)
        try:
            part = torch.empty((lib.conv_fwd_row_blocks(desc) + 64) * 2 * Co, device=d)
            t1 = timeit(lambda: ops.conv_fwd(x, w, None, y, part, desc, st), 20)
            t2 = timeit(lambda: ops.conv_dgrad(dy, w, dx, desc, st), 20)
            if ref is None:
                ref = (y.clone(), dx.clone())
            else:
                assert float((y - ref[0]).abs().max()) < 1e-3 * float(ref[0].abs().max()) and float((dx - ref[1]).abs().max()) < 1e-3 * float(ref[1].abs().max())
        finally:
            lib.set_streamk_slots(0)
        cells.append('%6.3f / %6.3f' % (t1, t2))
    print('%-30s %s' % (str((B, H, W, Ci, Co, k)), '   '.join(cells)), flush=True)
