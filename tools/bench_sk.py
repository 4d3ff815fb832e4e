"""Micro-benchmark: forward / dgrad of the low-tile-count GEMMs of the step (token-side Linear layers, 1x1 convolutions) with the
planner's schedule against a forced stream-K schedule over n workgroups (mmi_set_streamk_slots)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
from mmidet_hip import lib, ops  # noqa: E402
from bench_conv import timeit  # noqa: E402

SHAPES = [(2048, 1, 1, 1024, 256, 1, 1), (2048, 1, 1, 256, 1024, 1, 1), (2048, 1, 1, 2048, 512, 1, 1), (2048, 1, 1, 512, 2048, 1, 1),
          (2048, 1, 1, 512, 512, 1, 1), (2048, 1, 1, 512, 1536, 1, 1), (2048, 1, 1, 1024, 1024, 1, 1), (2048, 1, 1, 1024, 3072, 1, 1),
          (2048, 1, 1, 256, 256, 1, 1), (2048, 1, 1, 256, 768, 1, 1), (2048, 1, 1, 128, 128, 1, 1), (2048, 1, 1, 128, 384, 1, 1),
          (16, 80, 80, 128, 128, 1, 1), (16, 40, 40, 256, 256, 1, 1), (16, 160, 160, 64, 64, 1, 1), (16, 20, 20, 512, 512, 1, 1)]


def main():
    d = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    slots = [0, 256, 512, 768]
    print('%-32s %-6s' % ('shape', 'dir') + ''.join('%10s' % ('planner' if s == 0 else 'sk%d' % s) for s in slots) + '   (TFLOP/s)')
    for (B, H, W, Ci, Co, k, s) in SHAPES:
        x = torch.randn(B, H, W, Ci, device=d)
        w = torch.randn(Co, k, k, Ci, device=d) * 0.05
        desc = ops._desc((B, H, W, Ci), Co, k, s, Ci, Co)
        y = torch.empty(B, desc.Ho, desc.Wo, Co, device=d)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        fl = 2.0 * B * desc.Ho * desc.Wo * Co * Ci * k * k
        res = {'fwd': [], 'dgrad': []}
        for n in slots:
            lib.set_streamk_slots(n)
            res['fwd'].append(fl / timeit(lambda: ops.conv_fwd(x, w, None, y, None, desc, st), 20) / 1e9)
            res['dgrad'].append(fl / timeit(lambda: ops.conv_dgrad(dy, w, dx, desc, st), 20) / 1e9)
        lib.set_streamk_slots(0)
        for dname, vals in res.items():
            print('%-32s %-6s' % (str((B, H, W, Ci, Co, k, s)), dname) + ''.join('%10.1f' % v for v in vals), flush=True)


if __name__ == '__main__':
    main()
