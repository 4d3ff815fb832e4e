"""Wave-quantisation probe: the same 3x3 128->128 conv at tile counts around the number of resident workgroup slots."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mmidet_hip import lib, ops  # noqa: E402
from bench_conv import timeit  # noqa: E402

d = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for rows128 in (256, 512, 640, 768, 769, 800, 896, 1024, 1025, 1280, 1536, 1537, 2048):
    H, W = rows128, 128
    x = torch.randn(1, H, W, C, device=d)
    w = torch.randn(C, 3, 3, C, device=d) * 0.05
    desc = ops._desc((1, H, W, C), C, 3, 1, C, C)
    y = torch.empty(1, H, W, C, device=d)
    part = torch.empty(lib.conv_fwd_row_blocks(desc) * 2 * C, device=d)
    fl = 2.0 * H * W * C * C * 9
    t1 = timeit(lambda: ops.conv_fwd(x, w, None, y, part, desc, st))
    t2 = timeit(lambda: ops.conv_dgrad(y, w, x, desc, st))
    print('m-tiles %5d  fwd %.3f ms %6.1f TF   dgrad %.3f ms %6.1f TF' % (rows128, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9), flush=True)
