"""Where does the HOST time of one eager training step go?  cProfile over a few steps of the BASELINE workload
(yolov5l two-stream-fourier, B=16, 640x640), top functions by own time.  The GPU runs ~123 ms per step; the host must stay
below that with margin, on one core, for eight ranks on a 16-core box (VERDICT r1 "weak" 12).

    python tools/host_profile.py [steps] > gpurun_out/host_profile.txt
"""
import cProfile
import io
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mmi-det_amd'))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    tiny = len(sys.argv) > 2 and sys.argv[2] == 'tiny'     # B=1, 64x64: the GPU work vanishes, the step time IS the host time
    bsz, size = (1, 64) if tiny else (16, 640)
    from mmidet_hip.train_step import TrainStep
    from models.yolo_test import Model
    dev = torch.device('cuda:0')
    cfg = bench.load_cfg('l_fourier')
    model = Model(cfg).to(dev).train()
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.1
    ts = TrainStep(model, cfg['nc'], size, bsz, accumulate=1)
    imgs, tg = bench.synth(bsz, size, cfg['nc'], dev, 100)
    for _ in range(3):
        ts.step(imgs, tg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step(imgs, tg)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('plain%s: host enqueue %.1f ms/step, wall %.1f ms/step' % (' (tiny batch: pure host cost)' if tiny else '', t_enq / steps * 1e3, t_all / steps * 1e3))
    if tiny:      # phase split with a device sync between phases (only meaningful when the GPU is not the bottleneck)
        import types
        from mmidet_hip import ops, fusion_ops as F2
        acc = [0.0, 0.0, 0.0, 0.0]
        for _ in range(steps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            F2.advance_seed(dev)
            rgb, ir = ops.u8_pair_to_nhwc(imgs)
            pred, comb = model(rgb, ir)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss, items = ts.compute_loss(pred, tg, comb.reshape(-1))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ops.DEFER_JOIN = True
            loss.sum().backward()
            ops.DEFER_JOIN = False
            ops.join_pending()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            ts._update()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for i, d in enumerate((t1 - t, t2 - t1, t3 - t2, t4 - t3)):
                acc[i] += d
        print('phases (ms/step): forward %.1f  loss %.1f  backward %.1f  optimizer+zero %.1f' % tuple(a / steps * 1e3 for a in acc))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        ts.step(imgs, tg)
    pr.disable()
    torch.cuda.synchronize()
    out = io.StringIO()
    st = pstats.Stats(pr, stream=out)
    st.sort_stats('tottime').print_stats(45)
    txt = out.getvalue()
    print('(per-call numbers below are inflated ~2x by the profiler itself; read the shares)\n')
    print('\n'.join(l for l in txt.splitlines() if l.strip())[:9000])
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(35)
    print('\n---- by cumulative time ----')
    print('\n'.join(l for l in out.getvalue().splitlines() if l.strip())[:7000])


if __name__ == '__main__':
    main()
