"""Micro-benchmark of the MFMA implicit-GEMM kernels on the yolov5l layer shapes.
Prints achieved TFLOP/s per shape for fwd / dgrad / wgrad (HIP events on the launch stream)."""
import sys
import os
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
from mmidet_hip import lib, ops  # noqa: E402

SHAPES = [  # B, H, W, Cin, Cout, k, s
    (16, 160, 160, 64, 64, 3, 1), (16, 80, 80, 128, 128, 3, 1), (16, 40, 40, 256, 256, 3, 1), (16, 20, 20, 512, 512, 3, 1),
    (16, 320, 320, 64, 128, 3, 2), (16, 160, 160, 128, 256, 3, 2), (16, 80, 80, 256, 512, 3, 2), (16, 40, 40, 512, 1024, 3, 2),
    (16, 160, 160, 128, 64, 1, 1), (16, 80, 80, 256, 128, 1, 1), (16, 40, 40, 512, 256, 1, 1), (16, 20, 20, 1024, 512, 1, 1),
    (16, 20, 20, 2048, 1024, 1, 1), (16, 320, 320, 12, 64, 3, 1), (2048, 1, 1, 1024, 4096, 1, 1), (2048, 1, 1, 4096, 1024, 1, 1),
    (16, 160, 160, 64, 64, 1, 1), (16, 80, 80, 128, 128, 1, 1), (16, 40, 40, 256, 256, 1, 1), (16, 20, 20, 512, 512, 1, 1),
    (2048, 1, 1, 128, 128, 1, 1), (2048, 1, 1, 256, 256, 1, 1), (2048, 1, 1, 512, 512, 1, 1), (2048, 1, 1, 1024, 1024, 1, 1),
]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    if os.environ.get('BENCH_UNI') is not None:   # A/B: 0 = the general cursor-based loaders everywhere
        lib.set_uniform_loaders(int(os.environ['BENCH_UNI']))
    d = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    print('%-34s %9s %9s %9s %9s   (TFLOP/s; ms)' % ('shape', 'fwd', 'dgrad', 'wgrad', 'wgrad+tab'))
    for (B, H, W, Ci, Co, k, s) in SHAPES:
        x = torch.randn(B, H, W, Ci, device=d)
        w = torch.randn(Co, k, k, Ci, device=d) * 0.05
        desc = ops._desc((B, H, W, Ci), Co, k, s, Ci, Co)
        y = torch.empty(B, desc.Ho, desc.Wo, Co, device=d)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        nb = lib.conv_wgrad_workspace(desc)
        ws = torch.zeros(max(nb // 4, 1), device=d)     # (arrival counters at its head: zero-filled once)
        part = torch.empty(lib.conv_fwd_row_blocks(desc) * 2 * Co, device=d)
        fl = 2.0 * B * desc.Ho * desc.Wo * Co * Ci * k * k
        t1 = timeit(lambda: ops.conv_fwd(x, w, None, y, part, desc, st))
        t2 = timeit(lambda: ops.conv_dgrad(dy, w, dx, desc, st))
        t3 = timeit(lambda: lib.conv_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, desc, st))
        tb = lib.conv_wgrad_table_bytes(desc)
        t4 = t3
        if tb:      # the same with the layer's precomputed pixel table
            tab = torch.empty(tb, dtype=torch.uint8, device=d)
            lib.conv_wgrad_table_build(tab.data_ptr(), desc, st)
            t4 = timeit(lambda: lib.conv_wgrad_tab(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, tab.data_ptr(), desc, st))
        print('%-34s %9.1f %9.1f %9.1f %9.1f   (%.3f %.3f %.3f %.3f)' % (str((B, H, W, Ci, Co, k, s)), fl / t1 / 1e9, fl / t2 / 1e9,
                                                                          fl / t3 / 1e9, fl / t4 / 1e9, t1, t2, t3, t4), flush=True)


if __name__ == '__main__':
    main()
