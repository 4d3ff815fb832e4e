"""Tile-variant sweep over the short-K GEMM shapes of the yolov5l step (1x1 convs, token projections): times every
forward/dgrad kernel variant with one workgroup per tile against the planner's own choice.  Calibrates plan_igemm."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mmidet_hip import lib, ops  # noqa: E402
from bench_conv import timeit  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout, k
    (16, 160, 160, 64, 64, 1), (16, 160, 160, 128, 64, 1), (16, 160, 160, 128, 128, 1), (16, 80, 80, 128, 128, 1),
    (16, 80, 80, 256, 128, 1), (16, 80, 80, 256, 256, 1), (16, 80, 80, 512, 128, 1), (16, 40, 40, 256, 256, 1),
    (16, 40, 40, 512, 256, 1), (16, 40, 40, 512, 512, 1), (16, 40, 40, 1024, 256, 1), (16, 20, 20, 512, 512, 1),
    (16, 20, 20, 1024, 512, 1), (16, 20, 20, 1024, 1024, 1), (16, 20, 20, 2048, 1024, 1),
    (2048, 1, 1, 128, 128, 1), (2048, 1, 1, 256, 256, 1), (2048, 1, 1, 512, 512, 1), (2048, 1, 1, 1024, 1024, 1),
    (2048, 1, 1, 256, 1024, 1), (2048, 1, 1, 1024, 256, 1), (2048, 1, 1, 512, 2048, 1), (2048, 1, 1, 2048, 512, 1),
    (2048, 1, 1, 1024, 4096, 1), (2048, 1, 1, 4096, 1024, 1),
    (16, 160, 160, 64, 64, 3), (16, 320, 320, 12, 64, 3),
]
VARIANTS = [(0, 0), (128, 128), (128, 64), (64, 64)]


def main():
    d = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    print('%-32s %s' % ('shape', '   '.join('%14s' % ('auto' if v == (0, 0) else '%dx%d' % v) for v in VARIANTS)), '  (fwd TF / dgrad TF)')
    for (B, H, W, Ci, Co, k) in SHAPES:
        x = torch.randn(B, H, W, Ci, device=d)
        w = torch.randn(Co, k, k, Ci, device=d) * 0.05
        desc = ops._desc((B, H, W, Ci), Co, k, 1, Ci, Co)
        y = torch.empty(B, desc.Ho, desc.Wo, Co, device=d)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        fl = 2.0 * B * desc.Ho * desc.Wo * Co * Ci * k * k
        cells = []
        for v in VARIANTS:
            if v[1] == 128 and (Co <= 64 or Ci <= 64):
                cells.append('%14s' % '-')
                continue
            lib.set_tile_override(*v)
            try:
                part = torch.empty((lib.conv_fwd_row_blocks(desc) + 64) * 2 * Co, device=d)
                t1 = timeit(lambda: ops.conv_fwd(x, w, None, y, part, desc, st), 20)
                t2 = timeit(lambda: ops.conv_dgrad(dy, w, dx, desc, st), 20)
            finally:
                lib.set_tile_override(0, 0)
            cells.append('%6.1f /%6.1f' % (fl / t1 / 1e9, fl / t2 / 1e9))
        print('%-32s %s' % (str((B, H, W, Ci, Co, k)), '   '.join(cells)), flush=True)


if __name__ == '__main__':
    main()
