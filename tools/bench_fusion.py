"""Micro-benchmark of the memory-bound fusion ops (SURVEY.md §8a rows 7, 8, 10, 12, 13 and the layout ops) at the BASELINE
shapes: achieved GB/s of ALGORITHMIC traffic (each input read once, each output written once) against the 8 TB/s HBM peak."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mmidet_hip import fusion_ops as F2, ops  # noqa: E402
from bench_conv import timeit  # noqa: E402

d = torch.device('cuda:0')
B = 16


def row(name, bytes_, fn, n=20):
    t = timeit(fn, n)
    print('%-58s %8.1f MB  %8.1f us  %7.0f GB/s  (%.2f of 8 TB/s)' % (name, bytes_ / 1e6, t * 1e3, bytes_ / t / 1e6, bytes_ / t / 8e9),
          flush=True)


def main():
    print('%-58s %11s %12s %13s' % ('op @ shape', 'algorithmic', 'time', 'achieved'))
    with torch.no_grad():
        for (h, c, tag) in ((160, 128, 'P2'), (80, 256, 'P3'), (40, 512, 'P4'), (20, 1024, 'P5')):
            a = torch.randn(B, h, h, c, device=d)
            b = torch.randn(B, h, h, c, device=d)
            nb = a.numel() * 4
            tok = F2.pool_tokens(a, b)
            row('avg-pool 8x8 of both streams -> tokens  %s (%d,%d,%d,%d)' % (tag, B, h, h, c), 2 * nb + tok.numel() * 4,
                lambda: F2.pool_tokens(a, b))
            t1, _ = F2.split_tokens(tok)
            row('bilinear 8x8->HxW + Add2                %s' % tag, 2 * nb + t1.numel() * 4, lambda: F2.upsample_add(a, t1))
            row('Add (rgb + ir)                          %s' % tag, 3 * nb, lambda: ops.add(a, b))
            if tag == 'P2':
                row('CBM + IGM statistics (8 moments, 3 pair sums, 3 histograms) P2', 2 * nb, lambda: F2.fusion_stats(a, b, tok))
        x = torch.randn(B, 20, 20, 512, device=d)
        row('SPP cascaded 5/9/13 max-pools -> concat buffer (16,20,20,512)', x.numel() * 4 * 5, lambda: ops.spp_pool(x))
        u8 = torch.randint(0, 256, (B, 6, 640, 640), dtype=torch.uint8, device=d)
        row('uint8 (16,6,640,640) -> 2 x fp32 NHWC /255', u8.numel() + 2 * B * 640 * 640 * 3 * 4, lambda: ops.u8_pair_to_nhwc(u8))
        img = torch.randn(B, 640, 640, 3, device=d)
        row('Focus space-to-depth (16,640,640,3)', 2 * img.numel() * 4, lambda: ops.space_to_depth(img))
        r = torch.randn(B, 640, 640, 24, device=d)
        fac, bias = torch.ones(24, 1, 1, 1, device=d), torch.zeros(24, device=d)
        row('CEM stencil bank r + sobel(r) (16,640,640,24)', 2 * r.numel() * 4, lambda: ops.sobel_add(r, fac, bias), 10)
        y5 = torch.randn(B, 80, 80, 33, device=d)
        row('Detect permute (16,80,80,33) -> (16,3,80,80,11)', 2 * y5.numel() * 4, lambda: ops.head_permute(y5, 3))


if __name__ == '__main__':
    main()
