"""Split-bf16 forward GEMM (mmi_set_gemm_precision(1)) against the exact fp32-MFMA form: error and speed per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mmidet_hip import lib, ops  # noqa: E402
from bench_conv import timeit  # noqa: E402

MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # 1 = two-term split (3 products), 2 = three-term (6 products)
d = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(16, 80, 80, 128, 128, 3, 1), (16, 40, 40, 256, 256, 3, 1), (16, 20, 20, 512, 512, 3, 1), (16, 160, 160, 64, 64, 3, 1),
          (16, 160, 160, 128, 256, 3, 2), (16, 80, 80, 128, 128, 1, 1), (16, 40, 40, 512, 256, 1, 1), (2048, 1, 1, 1024, 4096, 1, 1),
          (1, 1536, 128, 256, 256, 3, 1)]
print('%-30s %8s %8s %5s %8s | %8s %8s %5s %8s | %8s %8s %5s %8s' % ('shape', 'fwd f32', 'fwd b3', 'x', 'err', 'dgr f32', 'dgr b3', 'x', 'err', 'wgr f32', 'wgr b3', 'x', 'err'))
for (B, H, W, Ci, Co, k, s) in SHAPES:
    x = torch.randn(B, H, W, Ci, device=d)
    w = torch.randn(Co, k, k, Ci, device=d) / (Ci * k * k) ** 0.5
    desc = ops._desc((B, H, W, Ci), Co, k, s, Ci, Co)
    fl = 2.0 * B * desc.Ho * desc.Wo * Co * Ci * k * k
    res = {}
    for mode in (0, MODE):
        lib.set_gemm_precision(mode)
        try:
            y = torch.empty(B, desc.Ho, desc.Wo, Co, device=d)
            part = torch.zeros((lib.conv_fwd_row_blocks(desc) + 64) * 2 * Co, device=d)
            t = timeit(lambda: ops.conv_fwd(x, w, None, y, part, desc, st), 20)
            dy = torch.randn(B, desc.Ho, desc.Wo, Co, device=d, generator=torch.Generator(device=d).manual_seed(1))
            dx = torch.empty_like(x)
            t2 = timeit(lambda: ops.conv_dgrad(dy, w, dx, desc, st), 20)
            dw = torch.empty_like(w)
            nb = lib.conv_wgrad_workspace(desc)
            ws = torch.zeros(max(nb // 4, 1), device=d)     # (arrival counters at its head: zero-filled once)
            t3 = timeit(lambda: lib.conv_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, desc, st), 20)
            res[mode] = (y.clone(), t, dx.clone(), t2, dw.clone(), t3)
        finally:
            lib.set_gemm_precision(0)
    e = float((res[MODE][0].double() - res[0][0].double()).norm() / res[0][0].double().norm())
    e2 = float((res[MODE][2].double() - res[0][2].double()).norm() / res[0][2].double().norm())
    e3 = float((res[MODE][4].double() - res[0][4].double()).norm() / res[0][4].double().norm())
    print('%-30s %8.1f %8.1f %5.2f %8.1e | %8.1f %8.1f %5.2f %8.1e | %8.1f %8.1f %5.2f %8.1e' % (
        str((B, H, W, Ci, Co, k, s)), fl / res[0][1] / 1e9, fl / res[MODE][1] / 1e9, res[0][1] / res[MODE][1], e,
        fl / res[0][3] / 1e9, fl / res[MODE][3] / 1e9, res[0][3] / res[MODE][3], e2,
        fl / res[0][5] / 1e9, fl / res[MODE][5] / 1e9, res[0][5] / res[MODE][5], e3), flush=True)
