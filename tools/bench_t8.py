"""Micro-benchmark: the three-term bf16 GEMM modes with the split in the kernel against pre-split (T8) operands
(include/mmidet_hip.h: mmi_gemm_operands_t8), beside the fp32-MFMA kernels.  TFLOP/s fp32-equivalent per shape and direction."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
from mmidet_hip import lib, ops  # noqa: E402
from bench_conv import timeit  # noqa: E402

SHAPES = [(16, 160, 160, 64, 64, 3, 1), (16, 80, 80, 128, 128, 3, 1), (16, 40, 40, 256, 256, 3, 1), (16, 20, 20, 512, 512, 3, 1),
          (16, 160, 160, 128, 256, 3, 2), (16, 80, 80, 256, 512, 3, 2),
          (16, 160, 160, 128, 64, 1, 1), (16, 80, 80, 256, 128, 1, 1), (16, 40, 40, 512, 256, 1, 1), (16, 20, 20, 1024, 512, 1, 1),
          (2048, 1, 1, 1024, 4096, 1, 1), (2048, 1, 1, 512, 512, 1, 1)]


def main():
    d = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    modes = [('fp32', 0, 0, 0), ('fp32/pf2', 0, 0, 1), ('x9', 3, 0, 0), ('x9/pf2', 3, 0, 1), ('x9+t8', 3, 3, 0), ('x9+t8/pf2', 3, 3, 1), ('x6', 2, 0, 0),
             ('x6/pf2', 2, 0, 1), ('x6+t8', 2, 3, 0), ('x6+t8/pf2', 2, 3, 1)]
    if os.environ.get('BENCH_MODES'):
        modes = [m for m in modes if m[0] in os.environ['BENCH_MODES'].split(',')]
    print('%-32s %-6s' % ('shape', 'dir') + ''.join('%10s' % m[0] for m in modes) + '   (TFLOP/s fp32-equivalent)')
    for (B, H, W, Ci, Co, k, s) in SHAPES:
        x = torch.randn(B, H, W, Ci, device=d)
        w = torch.randn(Co, k, k, Ci, device=d) * 0.05
        desc = ops._desc((B, H, W, Ci), Co, k, s, Ci, Co)
        y = torch.empty(B, desc.Ho, desc.Wo, Co, device=d)
        dy = torch.randn_like(y)
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        x8, w8, dy8 = ops.t8_image(x), ops.t8_image(w.reshape(Co, -1)), ops.t8_image(dy)
        tb = lib.conv_wgrad_table_bytes(desc)
        tab = None
        if tb:
            tab = torch.empty(tb, dtype=torch.uint8, device=d)
            lib.conv_wgrad_table_build(tab.data_ptr(), desc, st)
        fl = 2.0 * B * desc.Ho * desc.Wo * Co * Ci * k * k
        res = {'fwd': [], 'dgrad': [], 'wgrad': []}
        for name, prec, t8, pf2 in modes:
            lib.set_gemm_precision(prec)
            lib.set_deep_prefetch(pf2)
            nb = lib.conv_wgrad_workspace(desc)
            ws = torch.zeros(max(nb // 4, 1), device=d)
            part = torch.empty(lib.conv_fwd_row_blocks(desc) * 2 * Co, device=d)

            def fwd():
                if t8:
                    lib.gemm_operands_t8(x8.data_ptr() if t8 & 2 else None, w8.data_ptr(), None, None)
                ops.conv_fwd(x, w, None, y, part, desc, st)

            def dgrad():
                if t8:
                    lib.gemm_operands_t8(dy8.data_ptr() if t8 & 2 else None, w8.data_ptr(), None, None)
                ops.conv_dgrad(dy, w, dx, desc, st)

            def wgrad():
                if t8 & 2:
                    lib.gemm_operands_t8(dy8.data_ptr(), x8.data_ptr(), None, None)
                lib.conv_wgrad_tab(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, tab.data_ptr() if tab is not None else None,
                                   desc, st)
            res['fwd'].append(fl / timeit(fwd) / 1e9)
            res['dgrad'].append(fl / timeit(dgrad) / 1e9)
            res['wgrad'].append(fl / timeit(wgrad) / 1e9)
            lib.set_gemm_precision(0)
            lib.set_deep_prefetch(0)
        for dname, vals in res.items():
            print('%-32s %-6s' % (str((B, H, W, Ci, Co, k, s)), dname) + ''.join('%10.1f' % v for v in vals), flush=True)


if __name__ == '__main__':
    main()
