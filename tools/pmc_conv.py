"""One conv shape, a handful of launches of fwd / dgrad / wgrad: the target of `rocprofv3 --pmc ...` runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
from mmidet_hip import lib, ops  # noqa: E402

B, H, W, Ci, Co, k, s = [int(v) for v in (sys.argv[1:8] if len(sys.argv) >= 8 else '16 80 80 128 128 3 1'.split())]
d = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, W, Ci, device=d)
w = torch.randn(Co, k, k, Ci, device=d) * 0.05
desc = ops._desc((B, H, W, Ci), Co, k, s, Ci, Co)
y = torch.empty(B, desc.Ho, desc.Wo, Co, device=d)
dy = torch.randn_like(y)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
nb = lib.conv_wgrad_workspace(desc)
ws = torch.zeros(max(nb // 4, 1), device=d)     # (arrival counters at its head: zero-filled once)
part = torch.empty((lib.conv_fwd_row_blocks(desc) + 64) * 2 * Co, device=d)
tb = lib.conv_wgrad_table_bytes(desc)        # round 2: the layer's precomputed pixel table (None: in-kernel builder)
tab = None
if tb and os.environ.get('MMIDET_WGRAD_TABLE', '1') != '0':
    tab = torch.empty(tb, dtype=torch.uint8, device=d)
    lib.conv_wgrad_table_build(tab.data_ptr(), desc, st)
# PMC_PREC = 0 / 2 / 3: GEMM arithmetic (mmi_set_gemm_precision); PMC_T8 = 1: pre-split operand images (mmi_gemm_operands_t8)
prec, t8 = int(os.environ.get('PMC_PREC', '0')), os.environ.get('PMC_T8', '0') == '1'
if prec:
    lib.set_gemm_precision(prec)
    nb = lib.conv_wgrad_workspace(desc)
    ws = torch.zeros(max(nb // 4, 1), device=d)
if t8:
    x8, w8, dy8 = ops.t8_image(x), ops.t8_image(w.reshape(Co, -1)), ops.t8_image(dy)
for _ in range(3):
    if t8:
        lib.gemm_operands_t8(x8.data_ptr(), w8.data_ptr(), None, None)
    ops.conv_fwd(x, w, None, y, part, desc, st)
    if t8:
        lib.gemm_operands_t8(dy8.data_ptr(), w8.data_ptr(), None, None)
    ops.conv_dgrad(dy, w, dx, desc, st)
    if t8:
        lib.gemm_operands_t8(dy8.data_ptr(), x8.data_ptr(), None, None)
    lib.conv_wgrad_tab(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, tab.data_ptr() if tab is not None else None, desc, st)
torch.cuda.synchronize()
print('flop per launch', 2.0 * B * desc.Ho * desc.Wo * Co * Ci * k * k)
