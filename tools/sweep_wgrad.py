"""Weight-gradient planner sweep: for the shapes where the step loses most wgrad time (1x1 convolutions, token projections, the
3x3 layers), time every tile variant x a ladder of split-K counts (mmi_set_wgrad_override) against the planner's own choice.
Prints per shape the planner's time, the best forced (tile, splits) and the ratio; calibrates wgrad_plan."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mmidet_hip import lib, ops  # noqa: E402
from bench_conv import timeit  # noqa: E402

SHAPES = [  # (B, H, W, Cin, Cout, k, s), launches per step
    ((16, 80, 80, 128, 128, 3, 1), 21), ((16, 40, 40, 256, 256, 3, 1), 24), ((16, 20, 20, 512, 512, 3, 1), 9),
    ((16, 160, 160, 64, 64, 3, 1), 6), ((16, 80, 80, 128, 128, 1, 1), 21), ((16, 40, 40, 256, 256, 1, 1), 24),
    ((16, 20, 20, 512, 512, 1, 1), 9), ((16, 160, 160, 64, 64, 1, 1), 6), ((16, 160, 160, 128, 64, 1, 1), 4),
    ((16, 40, 40, 512, 256, 1, 1), 7), ((16, 20, 20, 1024, 512, 1, 1), 9), ((16, 80, 80, 256, 128, 1, 1), 4),
    ((2048, 1, 1, 1024, 1024, 1, 1), 32), ((2048, 1, 1, 512, 512, 1, 1), 32), ((2048, 1, 1, 256, 256, 1, 1), 32),
    ((2048, 1, 1, 128, 128, 1, 1), 32), ((2048, 1, 1, 1024, 4096, 1, 1), 8), ((2048, 1, 1, 4096, 1024, 1, 1), 8),
    ((2048, 1, 1, 1024, 3072, 1, 1), 8), ((2048, 1, 1, 512, 2048, 1, 1), 8), ((2048, 1, 1, 2048, 512, 1, 1), 8),
    ((16, 160, 160, 128, 128, 1, 1), 2), ((16, 80, 80, 256, 256, 1, 1), 3), ((16, 40, 40, 512, 512, 1, 1), 4),
    ((16, 20, 20, 1024, 1024, 1, 1), 3), ((16, 20, 20, 2048, 1024, 1, 1), 2), ((16, 40, 40, 1024, 256, 1, 1), 2),
    ((16, 80, 80, 512, 128, 1, 1), 2),
]
TILES = [(128, 128), (128, 64), (64, 128), (64, 64)]
LADDER = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256]


def main():
    d = torch.device('cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    tot_auto = tot_best = 0.0
    for (B, H, W, Ci, Co, k, s), per_step in SHAPES:
        x = torch.randn(B, H, W, Ci, device=d)
        w = torch.randn(Co, k, k, Ci, device=d) * 0.05
        desc = ops._desc((B, H, W, Ci), Co, k, s, Ci, Co)
        dy = torch.randn(B, desc.Ho, desc.Wo, Co, device=d)
        dw = torch.empty_like(w)

        def run():
            nb = lib.conv_wgrad_workspace(desc)
            ws = torch.zeros(max(nb // 4, 1) + 4096, device=d)
            tb = lib.conv_wgrad_table_bytes(desc)
            tab = None
            if tb:
                tab = torch.empty(tb, dtype=torch.uint8, device=d)
                lib.conv_wgrad_table_build(tab.data_ptr(), desc, st)
            return timeit(lambda: lib.conv_wgrad_tab(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb,
                                                     tab.data_ptr() if tab is not None else None, desc, st), 20)
        lib.set_wgrad_override(0, 0, 0)
        t_auto = run()
        ref = dw.clone()
        best = (t_auto, 'auto')
        mpix = B * desc.Ho * desc.Wo
        for bm, bn in TILES:
            if (bm == 128 and Co <= 64) or (bn == 128 and k * k * Ci <= 64):
                continue
            for sp in LADDER:
                if sp * 32 > mpix or (Co * k * k * Ci * 4 * sp) > (1 << 30):
                    continue
                lib.set_wgrad_override(bm, bn, sp)
                try:
                    t = run()
                except Exception:
                    continue
                if t < best[0]:
                    err = float((dw - ref).abs().max() / (ref.abs().max() + 1e-30))
                    if err < 1e-3:
                        best = (t, '%dx%d splits %d' % (bm, bn, sp))
        lib.set_wgrad_override(0, 0, 0)
        fl = 2.0 * mpix * Co * Ci * k * k
        tot_auto += t_auto * per_step
        tot_best += best[0] * per_step
        print('%-34s x%-3d auto %.3f ms (%5.1f TF)   best %.3f ms (%5.1f TF)  %-22s  %.2fx' % (
            str((B, H, W, Ci, Co, k, s)), per_step, t_auto, fl / t_auto / 1e9, best[0], fl / best[0] / 1e9, best[1], t_auto / best[0]), flush=True)
    print('sum over the step: auto %.2f ms, best %.2f ms' % (tot_auto, tot_best))


if __name__ == '__main__':
    main()
