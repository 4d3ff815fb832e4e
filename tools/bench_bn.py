"""Micro-benchmark of the BatchNorm/activation streaming kernels on the yolov5l map sizes: achieved GB/s of algorithmic
traffic (fwd: read z + write y; bwd reduce: read z, dout; bwd apply: read z, dout + write dz) against the 8 TB/s HBM peak."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
from mmidet_hip import lib  # noqa: E402
from bench_conv import timeit  # noqa: E402

SHAPES = [(16 * 320 * 320, 64), (16 * 160 * 160, 64), (16 * 160 * 160, 128), (16 * 80 * 80, 128), (16 * 80 * 80, 256),
          (16 * 40 * 40, 256), (16 * 40 * 40, 512), (16 * 20 * 20, 512), (16 * 20 * 20, 1024)]


def main():
    d = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    print('%-22s %12s %12s %12s %12s  (GB/s; us)' % ('rows x C', 'fwd', 'fwd+res', 'bwd_reduce', 'bwd_apply'))
    for rows, c in SHAPES:
        z = torch.randn(rows, c, device=d)
        res = torch.randn(rows, c, device=d)
        out = torch.empty_like(z)
        dout = torch.randn_like(z)
        dz = torch.empty_like(z)
        mi = torch.cat([torch.zeros(c, device=d), torch.ones(c, device=d)])
        g, b = torch.ones(c, device=d), torch.zeros(c, device=d)
        dg, db = torch.empty(c, device=d), torch.empty(c, device=d)
        nparts = lib.bn_bwd_parts(rows)
        part = torch.empty(nparts * 2 * c, device=d)
        nb = rows * c * 4
        t1 = timeit(lambda: lib.bn_act_fwd(z.data_ptr(), c, mi.data_ptr(), g.data_ptr(), b.data_ptr(), None, 0,
                                           out.data_ptr(), c, rows, c, lib.ACT_SILU, st), 20)
        t2 = timeit(lambda: lib.bn_act_fwd(z.data_ptr(), c, mi.data_ptr(), g.data_ptr(), b.data_ptr(), res.data_ptr(), c,
                                           out.data_ptr(), c, rows, c, lib.ACT_SILU, st), 20)
        t3 = timeit(lambda: lib.bn_act_bwd_reduce(z.data_ptr(), c, dout.data_ptr(), c, mi.data_ptr(), g.data_ptr(),
                                                  b.data_ptr(), part.data_ptr(), rows, c, lib.ACT_SILU, st), 20)
        t4 = timeit(lambda: lib.bn_act_bwd_apply(z.data_ptr(), c, dout.data_ptr(), c, mi.data_ptr(), g.data_ptr(),
                                                 b.data_ptr(), part.data_ptr(), nparts, dz.data_ptr(), c, dg.data_ptr(),
                                                 db.data_ptr(), rows, c, lib.ACT_SILU, 0, st), 20)
        print('%-22s %12.0f %12.0f %12.0f %12.0f  (%.1f %.1f %.1f %.1f)' % (
            '%d x %d' % (rows, c), 2 * nb / t1 / 1e6, 3 * nb / t2 / 1e6, 2 * nb / t3 / 1e6, 3 * nb / t4 / 1e6,
            t1 * 1e3, t2 * 1e3, t3 * 1e3, t4 * 1e3), flush=True)


if __name__ == '__main__':
    main()
